#!/bin/bash
# Run on the GPU box: kernel trace of the data-parallel step with its gradient reducer on (one rank over RCCL),
# reduced to the per-kernel breakdown of the last step.  usage: tools/dp_step_profile.sh <tag> [extra bench args]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp MI355_FORCE_REDUCER=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --prof-steps 0 "$@" > $out/bench.log 2>&1
trace=$(find $out -name '*kernel_trace.csv' | head -1)
{
  echo "# MI355_FORCE_REDUCER=1 (1 rank, RCCL) rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --prof-steps 0 $*"
  grep -o '{"metric.*' $out/bench.log
  echo
  echo "# last step: window between its optimizer launch and the previous one"
  python tools/graph_step_breakdown.py $trace 70 0 || true
} > $out/summary.txt
rm -f $trace
find $out -name '*.db' -delete
tail -3 $out/summary.txt
