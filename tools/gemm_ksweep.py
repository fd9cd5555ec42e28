"""Fixed cost vs per-k-tile cost of the GEMM kernel: sweep K at fixed M, N."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
d = "cuda:0"


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M, N in ((131072, 2560), (131072, 320), (32768, 2560), (32768, 320)):
    for K in (64, 128, 320, 640, 1280):
        x = torch.randn(M, K, device=d).to(torch.bfloat16)
        w = (torch.randn(N, K, device=d) * K ** -0.5).to(torch.bfloat16)
        y = torch.empty(M, N, device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(x, w, out=y))
        print(f"M{M} N{N} K{K}: {t:7.1f} us  {2.0*M*N*K/t/1e6:6.0f} TF  out {M*N*2/t/1e6:5.2f} TB/s")
