#!/bin/bash
# train step and sampling loop, library ab_tmp/libbase.so against the one in the tree, interleaved on one box; then the kernel tests
cd $GRAFT_REPO_ROOT
TAGS="base new" BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab_step.sh 2>&1 | tee gpurun_out/r3_ab_libs.log
for rep in 1 2; do
  for tag in base new; do
    unset MI355_DENOISER_LIB; [ $tag = new ] || export MI355_DENOISER_LIB=$GRAFT_REPO_ROOT/ab_tmp/lib$tag.so
    timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 20 2>/dev/null | grep '"hip_graph": true' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('sampler $tag', round(d['ms_per_step'], 3), 'ms', round(d['steps_per_s'], 2), 'steps/s')" || exit 1
  done
done 2>&1 | tee -a gpurun_out/r3_ab_libs.log
unset MI355_DENOISER_LIB
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_parity_shapes_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_ab_tests.log 2>&1; tail -3 gpurun_out/r3_ab_tests.log
