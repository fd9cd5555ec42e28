"""Fused GEGLU projection GEMM against the plain GEMM of the same shape, and against GEMM + separate gate kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd._lib import lib, check
d = "cuda:0"


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M, C in ((131072, 320), (32768, 640), (8192, 1280), (32768, 320), (8192, 640), (2048, 1280)):
    x = torch.randn(M, C, device=d).to(torch.bfloat16)
    w = (torch.randn(8 * C, C, device=d) * C ** -0.5).to(torch.bfloat16)
    b = torch.zeros(8 * C, device=d)
    y = torch.empty(M, 8 * C, device=d, dtype=torch.bfloat16)
    o = torch.empty(M, 4 * C, device=d, dtype=torch.bfloat16)
    t_f = timeit(lambda: ops.gemm_geglu(x, w, b))
    t_g = timeit(lambda: ops.gemm(x, w, bias=b, out=y))
    t_e = timeit(lambda: check(lib.mi355_geglu_fwd(y.data_ptr(), o.data_ptr(), M, 4 * C, ops._stream()), "geglu"))
    fl = 2.0 * M * C * 8 * C
    print(f"M{M} C{C}: fused {t_f:7.1f} us ({fl/t_f/1e6:6.0f} TF) | gemm {t_g:7.1f} us ({fl/t_g/1e6:6.0f} TF) + gate {t_e:6.1f} us = {t_g+t_e:7.1f}")
