#!/bin/bash
# per-family event-timed milliseconds of the train step under several environment settings: tools/ab_fam.sh "A=1 B=2" "A=3" ...
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python -u bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 1 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['roofline']['families']
print('$cfg', round(d['ms_per_step'], 3), 'ms', {k: (round(v['ms_per_step'], 2), int(v['launches'])) for k, v in f.items()})" || exit 1
done
