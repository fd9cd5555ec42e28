#!/bin/bash
# chunk walk of the GEMM epilogue with a fixed lane group per row, in every kernel: train step and sampling loop, old vs new library
cd $GRAFT_REPO_ROOT
TAGS="lnfold new" BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab_step.sh 2>&1 | tee gpurun_out/r3_rowmap_ab.log
for rep in 1 2; do
  for tag in lnfold new; do
    unset MI355_DENOISER_LIB; [ $tag = new ] || export MI355_DENOISER_LIB=$GRAFT_REPO_ROOT/ab_tmp/lib$tag.so
    timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 20 2>/dev/null | grep '"hip_graph": true' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('sampler $tag', round(d['ms_per_step'], 3), 'ms', round(d['steps_per_s'], 2), 'steps/s')" || exit 1
  done
done 2>&1 | tee -a gpurun_out/r3_rowmap_ab.log
