"""What the epilogue operands cost on top of a plain launch: 3x3 convolutions and dense GEMMs of the denoiser with
bias / residual / fp32 output, stand-alone timing (operands stay cache-resident between iterations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from tools.gemm_bench import timeit, d, BF

for (B, H, C, Co) in [(32, 32, 320, 320), (32, 16, 640, 640), (32, 8, 1280, 1280)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    b = torch.randn(Co, device=d)
    r = torch.randn(B, H, H, Co, device=d).to(BF)
    row = []
    for name, kw in (("plain", {}), ("bias", {"bias": b}), ("res", {"residual": r}), ("bias+res", {"bias": b, "residual": r}),
                     ("f32out", {"out_fp32": True})):
        t = timeit(lambda: ops.conv2d(x, w, Co, 3, **kw), iters=50)
        row.append(f"{name}: {t*1e6:6.1f} us")
    print(f"conv B{B} {H}x{H} {C}->{Co}  " + "  ".join(row))
for (M, K, N) in [(32768, 320, 320), (32768, 1280, 320), (8192, 640, 640), (8192, 2560, 640), (2048, 1280, 1280), (2048, 5120, 1280)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    b = torch.randn(N, device=d)
    r = torch.randn(M, N, device=d).to(BF)
    row = []
    for name, kw in (("plain", {}), ("bias", {"bias": b}), ("res", {"residual": r}), ("bias+res", {"bias": b, "residual": r}),
                     ("f32out", {"out_fp32": True})):
        t = timeit(lambda: ops.gemm(x, w, **kw), iters=50)
        row.append(f"{name}: {t*1e6:6.1f} us")
    print(f"gemm M{M} K{K} N{N}  " + "  ".join(row))
