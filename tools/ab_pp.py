"""A/B of the 256x320 ping-pong GEMM kernel against the kernels it replaces (mi355_tune_set(1, 1) switches it off), on the
large-grid shapes of the 512^2 sampling step (UNet batch 32) and of the train step, interleaved rounds in one process;
the two results are compared element-wise (different summation order: not bit-equal).  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd._lib import lib

d = torch.device("cuda:0")
BF = torch.bfloat16
# optional: ab_pp.py <key> <v0,v1,...> compares values of another tuning key (the kernel's own variants) instead
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 1
VALS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


cases = []
for (B, H, C, Co) in [(32, 64, 320, 320), (32, 64, 640, 320), (32, 64, 960, 320), (32, 32, 640, 640), (32, 32, 1280, 640),
                      (32, 32, 320, 640), (32, 64, 320, 640)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    cases.append((f"conv3 B{B} {H}x{H} {C}->{Co}", 2.0 * B * H * H * Co * 9 * C, lambda x=x, w=w, Co=Co: ops.conv2d(x, w, Co, 3)))
for (M, K, N) in [(131072, 320, 320), (131072, 320, 960), (131072, 1280, 320), (32768, 640, 640), (32768, 640, 1920),
                  (32768, 2560, 640), (32768, 320, 960), (32768, 320, 1280), (32768, 320, 640), (65536, 8192, 1280)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    cases.append((f"gemm M{M} K{K} N{N}", 2.0 * M * N * K, lambda x=x, w=w: ops.gemm(x, w)))
for (M, K, F) in [(131072, 320, 1280), (32768, 640, 2560), (32768, 320, 1280)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(2 * F, K, device=d) * K ** -0.5).to(BF)
    b = torch.randn(2 * F, device=d)
    cases.append((f"geglu M{M} K{K} F{F}", 4.0 * M * F * K, lambda x=x, w=w, b=b: ops.gemm_geglu(x, w, b)))

tot = {v: 0.0 for v in VALS}
for name, fl, fn in cases:
    best = {v: 1e9 for v in VALS}
    outs = {}
    for rnd in range(4):
        for v in VALS:
            lib.mi355_tune_set(KEY, v)
            best[v] = min(best[v], timeit(fn))
            if rnd == 0:
                o = fn()
                outs[v] = (o[0] if isinstance(o, tuple) else o).float().clone()
    err = max(float((outs[VALS[0]] - outs[v]).abs().max() / outs[VALS[0]].abs().max()) for v in VALS[1:])
    for v in VALS:
        tot[v] += best[v]
    print(f"{name:34s} " + " | ".join(f"[{v}] {best[v]*1e6:8.1f} us {fl/best[v]/1e12:6.0f} TF" for v in VALS) +
          f"  max|diff|/max {err:.1e}", flush=True)
lib.mi355_tune_set(KEY, 0)
print(f"key {KEY} (key 1: 0 = ping-pong kernel on, 1 = off) sum:", {v: f"{t*1e6:.1f} us" for v, t in tot.items()})
