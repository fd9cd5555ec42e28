#!/bin/bash
# Cost side of the weight-gradient flush size in the data-parallel step (1 rank over RCCL, forward graph): run on the GPU box
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MI355_FORCE_REDUCER=1
p=29520
for rep in 1 2; do for f in 4096 2048 8192 1024; do
  p=$((p+1))
  MASTER_PORT=$p MI355_WGRAD_FLUSH=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('flush $f', round(d['ms_per_step'],3), 'ms', d['config']['forward_hip_graph'])" || exit 1
done; done
