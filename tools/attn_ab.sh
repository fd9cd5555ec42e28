#!/bin/bash
# A/B of attention kernels: per-kernel averages from rocprofv3 for one shape at a time.  TAGS="base new x": "new" is the
# library in the tree, any other tag T loads ab_tmp/libT.so (built by tools/attn_build_variant.sh T <flags>, or a copy of
# an older build); SHAPES="B H N D[ T];..." (default: the train step's and the 512^2 sampler's level-0 shapes).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_ab
mkdir -p $O
for tag in ${TAGS:-base new}; do
  if [ $tag = new ]; then unset MI355_DENOISER_LIB; else export MI355_DENOISER_LIB=$R/ab_tmp/lib$tag.so; fi
  IFS=';' read -ra SH <<< "${SHAPES:-32 8 1024 40;32 8 1024 40 77;32 8 256 80;32 8 4096 40}"
  for shape in "${SH[@]}"; do
    n=$(echo $shape | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/${tag}_$n -o p -- python3 $R/tools/attn_bench.py $shape > $O/${tag}_$n.log 2>&1
    echo "== $tag $shape"; tail -1 $O/${tag}_$n.log
    python3 $R/tools/rocpd_kernels.py $O/${tag}_$n attn_        # per-kernel averages from the run's rocpd database
  done
done
