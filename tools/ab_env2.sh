#!/bin/bash
# A/B of several environment settings on ONE box: tools/ab_env2.sh "A=1 B=2" "A=3" ...  (each argument is one setting)
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$cfg', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')" || exit 1
  done
done
