#!/bin/bash
# sampling loop with the attention of the folded path on the plain kernel (tune 0 1) and on the loop without a running
# maximum (default); interleaved on ONE box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for t in "0 1" "0 0"; do
    echo "tune $t: $(timeout -k 10 300 python tools/sampler_bench.py --mode sampler --steps 30 --tune $t 2>/dev/null | grep '"hip_graph": true' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', round(d['steps_per_s'],2), 'steps/s')")" || exit 1
  done
done
