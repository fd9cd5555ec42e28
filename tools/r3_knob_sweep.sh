#!/bin/bash
# launcher thresholds on ONE box, train step and sampling step: the fill a grid of 256 x 320 tiles must reach for the
# ping-pong kernel (MI355_GEMM_PP_FILL, percent) and the tile count from which the 256 x 160 kernel is taken (MI355_GEMM_WIDE_MIN)
cd $GRAFT_REPO_ROOT
for cfg in "MI355_GEMM_PP_FILL=80" "MI355_GEMM_PP_FILL=60" "MI355_GEMM_PP_FILL=70" "MI355_GEMM_PP_FILL=90" "MI355_GEMM_WIDE_MIN=384" "MI355_GEMM_WIDE_MIN=768" "MI355_GEMM_PP_FILL=80"; do
  t=$(env $cfg timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 0 2>/dev/null | grep -o '{"metric.*' | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))") || exit 1
  s=$(env $cfg timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 30 2>/dev/null | grep '"hip_graph": true' | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))") || exit 1
  echo "$cfg train $t ms sampler $s ms"
done
