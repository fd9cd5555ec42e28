#!/bin/bash
# the round-2 tree (ab_tmp/r2tree) against the current tree on ONE box: train step with event-timed families, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for tree in ab_tmp/r2tree .; do
    (cd $tree && timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 2 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$tree', round(d['ms_per_step'], 2), 'ms', {k: (round(v['ms_per_step'], 2), int(v['launches'])) for k, v in d['roofline']['families'].items()}, round(d['roofline']['measured_on_this_device']['mfma_bf16_random_operands_TFLOPs']))") || exit 1
  done
done
