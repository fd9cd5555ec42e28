#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_parity_shapes_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tap_tests.log 2>&1 || { tail -30 gpurun_out/r3_tap_tests.log; exit 1; }
tail -2 gpurun_out/r3_tap_tests.log
TAGS="MI355_CONV_TAP_INNER=0 new" BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab_step.sh 2>&1 | tee gpurun_out/r3_tap_ab.log
for rep in 1 2; do
  for cfg in MI355_CONV_TAP_INNER=0 MI355_CONV_TAP_INNER=1; do
    env $cfg timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 20 2>/dev/null | grep '"hip_graph": true' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('sampler $cfg', round(d['ms_per_step'], 3), 'ms', round(d['steps_per_s'], 2), 'steps/s')" || exit 1
  done
done 2>&1 | tee -a gpurun_out/r3_tap_ab.log
