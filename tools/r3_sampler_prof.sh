#!/bin/bash
# per-kernel totals of the sampling loop with a set of environment switches: tools/r3_sampler_prof.sh <tag> [VAR=value ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/prof_sampler_$tag; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/sampler_bench.py --mode sampler --steps 20 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
{ grep -o '{"mode.*' $out/bench.log; python tools/prof_summary.py $out 46 40; } > $out/summary.txt
find $out -name '*kernel_trace.csv' -delete; find $out -name '*.db' -delete
head -24 $out/summary.txt | cut -c1-170
