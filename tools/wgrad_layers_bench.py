"""One grouped weight-gradient launch per layer shape of the train step (batch 32), timed with events: microseconds and
TFLOP/s per shape.  A lone job is bounded below by one workgroup's walk over its 8 192 pixels (~100 us) and fills the
chip only when it has at least 256 tiles: the figures show which layers rely on the grouping, not the in-step cost.
usage: wgrad_layers_bench.py   (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
B = 32
SHAPES = [(32, 320, 320, 3), (32, 640, 320, 3), (32, 960, 320, 3), (16, 640, 640, 3), (16, 1280, 640, 3), (16, 1920, 640, 3),
          (8, 1280, 1280, 3), (8, 2560, 1280, 3), (4, 1280, 1280, 3), (4, 2560, 1280, 3),
          (32, 320, 2560, 1), (32, 1280, 320, 1), (32, 320, 320, 1), (16, 640, 5120, 1), (16, 2560, 640, 1), (16, 640, 640, 1),
          (8, 1280, 10240, 1), (8, 5120, 1280, 1), (8, 1280, 1280, 1)]
d = torch.device("cuda:0")
q = ops.WgradQueue(d, flush_blocks=1 << 30)
tot = 0.0
for (H, ci, co, k) in SHAPES:
    x = torch.randn(B, H, H, ci, device=d).to(torch.bfloat16)
    dy = torch.randn(B, H, H, co, device=d).to(torch.bfloat16)
    dw = torch.zeros(co, k, k, ci, device=d)
    best = 1e9
    for rep in range(6):
        q.push(dy, x, dw, None, B, H, H, ci, co, k, 1, False, H, H)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        q.flush()
        e1.record()
        torch.cuda.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1) * 1e-3)
    fl = 2.0 * B * H * H * ci * co * k * k
    tot += best
    print(f"{H:2d}x{H:<2d} {ci:5d}->{co:<5d} k{k}: {best*1e6:7.1f} us {fl/best/1e12:6.0f} TF", flush=True)
print(f"sum {tot*1e6:.1f} us")
