#!/bin/bash
# Data-parallel step at one rank over RCCL (forward graph) under several environment settings: tools/ab_dp_env.sh "A=1" "B=2" ...
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MI355_FORCE_REDUCER=1
p=29540
for rep in 1 2; do for cfg in "$@"; do
  p=$((p+1))
  env MASTER_PORT=$p $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), 'ms', round(d['value'],1), d['config']['forward_hip_graph'], d['config']['final_loss'], d['config']['final_grad_norm'])" || exit 1
done; done
