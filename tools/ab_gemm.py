"""A/B of contraction-kernel variants inside one process: the shapes of the train step that run at one workgroup per CU,
timed in interleaved rounds with mi355_tune_set(key, 0 | 1); results of the two variants compared bit for bit.
usage: ab_gemm.py [key]   (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd._lib import lib

d = torch.device("cuda:0")
BF = torch.bfloat16
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 0
VALS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]


def timeit(fn, iters=20):
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
for (B, H, C, Co) in [(32, 32, 320, 320), (32, 16, 640, 640), (32, 16, 1280, 640), (32, 8, 1280, 1280), (32, 8, 2560, 1280),
                      (32, 4, 1280, 1280), (32, 4, 2560, 1280)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    cases.append((f"conv3 B{B} {H}x{H} {C}->{Co}", 2.0 * B * H * H * Co * 9 * C, lambda x=x, w=w, Co=Co: ops.conv2d(x, w, Co, 3)))
for (M, K, N) in [(2048, 1280, 1280), (8192, 640, 640), (32768, 320, 320), (512, 1280, 1280), (2048, 5120, 1280),
                  (8192, 2560, 640), (2464, 768, 2560), (2048, 1280, 3840), (8192, 640, 1920), (2048, 10240, 1280),
                  (32768, 320, 960), (32768, 1280, 320)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    cases.append((f"gemm M{M} K{K} N{N}", 2.0 * M * N * K, lambda x=x, w=w: ops.gemm(x, w)))

tot = {v: 0.0 for v in VALS}
for name, fl, fn in cases:
    best = {v: 1e9 for v in VALS}
    outs = {}
    for rnd in range(5):
        for v in VALS:
            lib.mi355_tune_set(KEY, v)
            best[v] = min(best[v], timeit(fn))
            if rnd == 0:
                outs[v] = fn().clone()
    same = all(torch.equal(outs[VALS[0]], outs[v]) for v in VALS[1:])
    for v in VALS:
        tot[v] += best[v]
    print(f"{name:34s} " + "  ".join(f"[{v}] {best[v]*1e6:7.1f} us {fl/best[v]/1e12:6.0f} TF" for v in VALS) + f"  equal={same}")
lib.mi355_tune_set(KEY, 0)
print("sum:", {v: f"{t*1e6:.1f} us" for v, t in tot.items()})
