#!/bin/bash
# rocprofv3 kernel stats of the auto-encoder benchmark (encode 32 x 256^2 four times, decode 16 x 512^2 four times)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/prof_vae_r3; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/vae_bench.py > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
cat $out/bench.log | tail -3
python tools/prof_summary.py $out 4 40 > $out/summary.txt
find $out -name '*kernel_trace.csv' -delete; find $out -name '*.db' -delete
head -45 $out/summary.txt | cut -c1-200
