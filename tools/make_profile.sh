#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace + stats of the default bench command, reduced to the summaries kept
# under profiles/.  usage: tools/make_profile.sh <tag> [extra bench args]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra "$@" > $out/bench.log 2>&1
stats=$(find $out -name '*kernel_stats.csv' | head -1)
trace=$(find $out -name '*kernel_trace.csv' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra $*   (1 x MI355X)"
  echo "# bench line of the same (profiled) run:"
  grep -o '{"metric.*' $out/bench.log
  echo
  echo "# per-kernel totals divided by the 13 graphed + 3 eager steps of the run (includes model init kernels and the"
  echo "# spin kernels that bench.py puts in front of its event-timed steps)"
  python tools/prof_summary.py $out 16 45
  echo
  echo "# the last graph replay: window between its AdamW launch and the previous one (tools/graph_step_breakdown.py)"
  python tools/graph_step_breakdown.py $trace 60 3 || true
  echo
  echo "# the last eager, event-timed step of the same run (what the bench line's roofline.families are measured on;"
  echo "# the window starts with the spin kernel that lets the host run ahead)"
  python tools/graph_step_breakdown.py $trace 60 0 || true
} > $out/summary.txt
cp $stats $out/kernel_stats.csv
rm -f $trace
find $out -name '*.db' -delete
tail -5 $out/summary.txt
