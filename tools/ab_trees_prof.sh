#!/bin/bash
# per-kernel totals of the train step in an earlier commit's tree (ab_tmp/base_tree, see ab_trees.sh) and in the current tree, on ONE box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for tree in ab_tmp/base_tree .; do
  tag=$(echo $tree | tr -c 'a-z0-9\n' '_')
  (cd $tree && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abprof_$tag -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --prof-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/abprof_$tag.log 2>&1) || exit 1
  f=$(find gpurun_out/abprof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f gpurun_out/abprof_$tag.csv
  find gpurun_out/abprof_$tag -type f ! -name '*kernel_stats.csv' -delete
done
