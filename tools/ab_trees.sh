#!/bin/bash
# an earlier commit checked out and built under ab_tmp/base_tree (git worktree add ab_tmp/base_tree <commit>; make) against
# the current tree on ONE box: train step with event-timed families, interleaved.  A switch inside one build cannot show what
# a change costs the code around it (register allocation, spills); two trees can.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for tree in ab_tmp/base_tree .; do
    (cd $tree && timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 2 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$tree', round(d['ms_per_step'], 2), 'ms', {k: (round(v['ms_per_step'], 2), int(v['launches'])) for k, v in d['roofline']['families'].items()}, round(d['roofline']['measured_on_this_device']['mfma_bf16_random_operands_TFLOPs']))") || exit 1
  done
done
