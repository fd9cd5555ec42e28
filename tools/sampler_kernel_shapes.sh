#!/bin/bash
# per (kernel, grid size) totals of the sampling step from a kernel trace: which launches of a kernel template cost what
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/sampler_shapes; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python tools/sampler_bench.py --mode sampler --steps 10 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
python3 - <<PY
import csv, collections, glob, re
f = glob.glob('$out/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    if not any(k in name for k in ('gemm', 'attn', 'gn_', 'splitk')): continue
    key = (name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']))
    a = acc[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = 26.0      # 2 x (3 warm-up + 10 timed)
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
with open('$out/shapes.txt', 'w') as o:
    for (name, gx, gy), (n, us) in rows[:60]:
        line = '%-52s wgs %6d x %-5d  calls/step %5.1f  avg %8.1f us  ms/step %6.3f' % (name[:52], gx, gy, n / steps, us / n, us / steps / 1e3)
        print(line); o.write(line + '\n')
PY
find $out -name '*.csv' -delete; find $out -name '*.db' -delete
