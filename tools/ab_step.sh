#!/bin/bash
# interleaved whole-step A/B on one box: TAGS="base new" - "new" is the library in the tree, tag T loads ab_tmp/libT.so;
# a tag of the form VAR=value runs the library in the tree with that environment variable set (a host-side switch)
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for tag in ${TAGS:-base new}; do
    unset MI355_DENOISER_LIB; ENVSET=""
    case $tag in new) ;; *=*) ENVSET=$tag ;; *) export MI355_DENOISER_LIB=$R/ab_tmp/lib$tag.so ;; esac
    env $ENVSET timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --prof-steps 0 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1))" || exit 1
  done
done
