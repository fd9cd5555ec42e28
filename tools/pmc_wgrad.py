"""One grouped weight-gradient launch per layer shape of the train step, in a fixed order, for a counter pass:
rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_wgrad.py     (the reducer below matches launches to shapes by order)
python tools/pmc_wgrad.py --reduce <dir>  prints fetched bytes (2 x FETCH_SIZE KiB) against the operands' size."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = 32
SHAPES = [(32, 320, 320, 3), (32, 640, 320, 3), (32, 960, 320, 3), (16, 640, 640, 3), (16, 1280, 640, 3), (16, 1920, 640, 3),
          (8, 1280, 1280, 3), (8, 2560, 1280, 3), (4, 1280, 1280, 3), (4, 2560, 1280, 3),
          (32, 320, 2560, 1), (32, 1280, 320, 1), (16, 640, 5120, 1), (16, 2560, 640, 1), (8, 1280, 10240, 1), (8, 5120, 1280, 1)]
if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    f = glob.glob(f"{sys.argv[2]}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "wgrad_grouped" in r["Kernel_Name"]]
    rows = rows[-len(SHAPES):]
    for (H, ci, co, k), r in zip(SHAPES, rows):
        alg = B * H * H * (ci + co) * 2
        got = float(r["Counter_Value"]) * 1024 * 2
        print(f"{H}x{H} {ci}->{co} k{k}: operands {alg/1e6:7.1f} MB, fetched {got/1e6:8.1f} MB = {got/alg:5.2f} x")
    sys.exit(0)
import torch
from image_generator_amd import ops
d = torch.device("cuda:0")
q = ops.WgradQueue(d, flush_blocks=1 << 30)
bufs = []
for (H, ci, co, k) in SHAPES:
    bufs.append((torch.randn(B, H, H, ci, device=d).to(torch.bfloat16), torch.randn(B, H, H, co, device=d).to(torch.bfloat16),
                 torch.zeros(co, k, k, ci, device=d)))
for rep in range(2):
    for (H, ci, co, k), (x, dy, dw) in zip(SHAPES, bufs):
        q.push(dy, x, dw, None, B, H, H, ci, co, k, 1, False, H, H)
        q.flush()
        torch.cuda.synchronize()
