#!/bin/bash
# A/B of library builds on ONE box (devices differ by several percent): tools/ab_bench.sh libA.so libB.so ...
for lib in "$@"; do
  for rep in 1 2; do
    MI355_DENOISER_LIB=$lib timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 2 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms', {k: (round(v['ms_per_step'], 2), round(v['achieved'], 1)) for k, v in d['roofline']['families'].items()})" || exit 1
  done
done
