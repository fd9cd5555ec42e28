#!/bin/bash
# A/B of environment settings on ONE box with the event-timed kernel families: tools/ab_env_fam.sh "A=1" "A=0" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 2 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$cfg', round(d['ms_per_step'], 2), 'ms', {k: round(v['ms_per_step'], 2) for k, v in d['roofline']['families'].items()}, round(d['roofline']['measured_on_this_device']['mfma_bf16_random_operands_TFLOPs']))" || exit 1
  done
done
