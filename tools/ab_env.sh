#!/bin/bash
# A/B of one environment switch on ONE box: tools/ab_env.sh VAR "0 1" [bench args]
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    env $var=$v timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 "$@" 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')" || exit 1
  done
done
