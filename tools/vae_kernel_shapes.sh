#!/bin/bash
# per (kernel, grid size) totals of the auto-encoder's encode pass (32 x 256^2), from a kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/vae_shapes; rm -rf $out; mkdir -p $out
cat > $out/run.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from image_generator_amd.vae import AutoencoderKL, VaeConfig
d = torch.device("cuda:0")
torch.manual_seed(0)
vae = AutoencoderKL(VaeConfig.sd_v1()).to(d)
x = torch.rand(32, 3, 256, 256, device=d) * 2 - 1
for _ in range(5):
    vae.encode(x).latent_dist.mode()
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $out/run.py > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
python3 - <<PY
import csv, collections, glob, re
f = glob.glob('$out/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    key = (name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']))
    a = acc[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = 5.0
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in acc.values()) / steps / 1e3
print('kernel time per encode: %.2f ms' % tot)
for (name, gx, gy), (n, us) in rows[:40]:
    print('%-60s wgs %6d x %-5d  calls %5.1f  avg %8.1f us  ms %6.3f' % (name[:60], gx, gy, n / steps, us / n, us / steps / 1e3))
PY
find $out -name '*.csv' -delete; find $out -name '*.db' -delete
