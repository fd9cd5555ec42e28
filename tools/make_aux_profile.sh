#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace + stats of the 512x512 sampling loop and of the LoRA fine-tune step
# (tools/sampler_bench.py), reduced to per-kernel totals per step.  usage: tools/make_aux_profile.sh <tag>
set -e
tag=$1
export TMPDIR=/tmp
for mode in sampler lora; do
  out=gpurun_out/prof_${mode}_$tag
  rm -rf $out; mkdir -p $out
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/sampler_bench.py --mode $mode --steps 20 > $out/bench.log 2>&1
  {
    echo "# rocprofv3 --kernel-trace --stats -- python tools/sampler_bench.py --mode $mode --steps 20   (1 x MI355X)"
    echo "# result lines of the same (profiled) run: eager loop, then the replayed HIP graph"
    grep -o '{"mode.*' $out/bench.log
    echo
    echo "# per-kernel totals divided by the 46 steps of the run (2 x (3 warm-up + 20 timed); includes model init kernels)"
    python tools/prof_summary.py $out 46 40
  } > $out/summary.txt
  find $out -name '*kernel_trace.csv' -delete
  find $out -name '*.db' -delete
  head -12 $out/summary.txt | cut -c1-220
done
