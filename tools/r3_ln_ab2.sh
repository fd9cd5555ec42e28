#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "folded" > gpurun_out/r3_ln_tests.log 2>&1 || { tail -30 gpurun_out/r3_ln_tests.log; exit 1; }
tail -2 gpurun_out/r3_ln_tests.log
for rep in 1 2; do
  for cfg in MI355_NO_LN_FOLD=1 MI355_NO_LN_FOLD=0; do
    env $cfg timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 20 2>/dev/null | grep '"hip_graph": true' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'], 3), 'ms', round(d['steps_per_s'], 2), 'steps/s')" || exit 1
  done
done 2>&1 | tee gpurun_out/r3_ln_ab.log
