#!/bin/bash
# round-3 check of the one-product text key/value projection: tests, then interleaved step A/B on this box
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "unet or attention or split" > gpurun_out/r3_kv_tests.log 2>&1 || { tail -30 gpurun_out/r3_kv_tests.log; exit 1; }
tail -3 gpurun_out/r3_kv_tests.log
TAGS="MI355_NO_CTX_KV_FUSION=1 new" BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab_step.sh 2>&1 | tee gpurun_out/r3_kv_ab.log
