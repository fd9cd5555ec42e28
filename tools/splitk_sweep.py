"""Times GEMM shapes of the denoiser with and without bias / residual; run once per MI355_GEMM_SK value to see
what the split-K choice costs on one GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from tools.gemm_bench import timeit, d, BF

for (M, K, N) in [(2048, 1280, 1280), (512, 1280, 1280), (8192, 640, 640), (2048, 5120, 1280)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    b = torch.randn(N, device=d)
    r = torch.randn(M, N, device=d).to(BF)
    row = []
    for kw in ({}, {"bias": b}, {"residual": r}, {"bias": b, "residual": r}):
        t = timeit(lambda: ops.gemm(x, w, **kw), iters=50)
        row.append(f"{'+'.join(kw) or 'plain':>14}: {t*1e6:6.1f} us")
    print(f"M{M} K{K} N{N} sk={os.environ.get('MI355_GEMM_SK', 'auto')}  " + "  ".join(row))
for (B, H, C, Co) in [(32, 4, 1280, 1280), (32, 8, 1280, 1280)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    t = timeit(lambda: ops.conv2d(x, w, Co, 3), iters=50)
    print(f"conv B{B} {H}x{H} {C}->{Co} sk={os.environ.get('MI355_GEMM_SK', 'auto')}: {t*1e6:7.1f} us {2.0*B*H*H*Co*9*C/t/1e12:7.1f} TF")
