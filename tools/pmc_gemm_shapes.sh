#!/bin/bash
# Run on the GPU box: per-shape fetched / written bytes of the contraction kernels (tools/pmc_gemm_shapes.py).  usage: tools/pmc_gemm_shapes.sh <tag>
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pmc_shapes_$1; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python tools/pmc_gemm_shapes.py > $out/$c.log 2>&1
  echo "$c pass done"
done
python tools/pmc_gemm_shapes.py --reduce $out > $out/gemm_fetch_per_shape.txt
find $out -name '*.db' -delete; find $out -name '*counter_collection.csv' -delete
cat $out/gemm_fetch_per_shape.txt
