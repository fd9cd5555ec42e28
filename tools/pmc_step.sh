#!/bin/bash
# Run on the GPU box: HBM-side traffic of every kernel of the eager train step, one rocprofv3 --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), reduced to profiles-ready JSON by tools/pmc_step_traffic.py.
# usage: tools/pmc_step.sh <tag>
set -e
tag=$1
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python bench.py --no-graph --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline --no-extra > $out/$c.log 2>&1
  echo "$c pass done"
done
python tools/pmc_step_traffic.py $out > $out/step_traffic.json
find $out -name '*.db' -delete
cat $out/step_traffic.json
