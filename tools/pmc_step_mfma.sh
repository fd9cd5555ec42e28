#!/bin/bash
# Run on the GPU box: matrix-pipe utilisation of every contraction kernel of the eager train step from one
# rocprofv3 --pmc pass, reduced by tools/pmc_step_mfma.py.  usage: tools/pmc_step_mfma.sh <tag>
set -e
tag=$1
out=gpurun_out/pmc_mfma_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pass -o p -- python bench.py --no-graph --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline --no-extra > $out/pass.log 2>&1
python tools/pmc_step_mfma.py $out > $out/step_mfma.json
find $out -name '*.db' -delete
find $out -name '*counter_collection.csv' -delete
cat $out/step_mfma.json
