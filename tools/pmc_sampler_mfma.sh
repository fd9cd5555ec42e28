#!/bin/bash
# Run on the GPU box: matrix-pipe utilisation of every contraction kernel of the 512 x 512 sampling step (16 images +
# guidance) from one rocprofv3 --pmc pass, windowed on the DDIM update launches.  usage: tools/pmc_sampler_mfma.sh <tag>
set -e
tag=$1
out=gpurun_out/pmc_sampler_mfma_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pass -o p -- python tools/sampler_bench.py --mode sampler --steps 3 > $out/pass.log 2>&1
python tools/pmc_step_mfma.py $out ddim_step_cfg_kernel "python tools/sampler_bench.py --mode sampler --steps 3" > $out/sampler_mfma.json
find $out -name '*.db' -delete
find $out -name '*counter_collection.csv' -delete
head -60 $out/sampler_mfma.json
