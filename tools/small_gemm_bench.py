"""Times the small GEMM / conv shapes of the denoiser; run once per value of an environment knob
(e.g. MI355_GEMM_NW8) to compare kernel variants on one GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from tools.gemm_bench import timeit, d, BF

for (M, K, N) in [(2048, 1280, 1280), (32768, 320, 320), (8192, 640, 640), (512, 1280, 1280), (2464, 768, 2560), (32768, 320, 960)]:
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    t = timeit(lambda: ops.gemm(x, w), iters=30)
    print(f"M{M} K{K} N{N}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
for (B, H, C, Co) in [(32, 4, 1280, 1280), (32, 8, 1280, 1280)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    t = timeit(lambda: ops.conv2d(x, w, Co, 3), iters=30)
    print(f"conv B{B} {H}x{H} {C}->{Co}: {t*1e6:7.1f} us {2.0*B*H*H*Co*9*C/t/1e12:7.1f} TF")
