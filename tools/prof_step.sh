# rocprofv3 kernel trace of a short bench.py run, per-kernel counts and averages (tools/rocpd_kernels.py); BENCH_ARGS, NLINES
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ab; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --prof-steps 0 ${BENCH_ARGS} > $O/p.log 2>&1
python tools/rocpd_kernels.py $O/p "" | head -${NLINES:-70}
