cd /tmp && export TMPDIR=/tmp
timeout -k 5 60 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_avail.txt 2>&1
grep -oE "SQ_[A-Z0-9_]+" $GRAFT_REPO_ROOT/gpurun_out/counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000
