"""Times LayerNorm / GroupNorm forward and backward on the denoiser's activation shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd.ops import LayerNormFn, GroupNormFn
from tools.gemm_bench import timeit, d, BF

for (M, C) in [(32768, 320), (8192, 640), (2048, 1280), (512, 1280)]:
    x = torch.randn(M, C, device=d).to(BF).requires_grad_(True)
    g = torch.ones(C, device=d, requires_grad=True); b = torch.zeros(C, device=d, requires_grad=True)
    y = LayerNormFn.apply(x, g, b, 1e-5, None)
    dy = torch.randn_like(y)
    tf = timeit(lambda: LayerNormFn.apply(x, g, b, 1e-5, None), iters=20)
    tb = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True), iters=20)
    print(f"LN M{M} C{C}: fwd {tf*1e6:6.1f} us ({M*C*4/tf/1e12:.2f} TB/s)  bwd {tb*1e6:6.1f} us ({M*C*6/tb/1e12:.2f} TB/s)")
for (B, H, C) in [(32, 32, 320), (32, 32, 640), (32, 32, 960), (32, 16, 640), (32, 16, 1280), (32, 8, 1280), (32, 8, 2560), (32, 4, 1280), (32, 4, 2560)]:
    x = torch.randn(B, H, H, C, device=d).to(BF).requires_grad_(True)
    g = torch.ones(C, device=d, requires_grad=True); b = torch.zeros(C, device=d, requires_grad=True)
    y = GroupNormFn.apply(x, g, b, 32, 1e-5, True, None)
    dy = torch.randn_like(y)
    n = B * H * H * C
    tf = timeit(lambda: GroupNormFn.apply(x, g, b, 32, 1e-5, True, None), iters=20)
    tb = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True), iters=20)
    print(f"GN B{B} {H}x{H} C{C}: fwd {tf*1e6:6.1f} us ({n*6/tf/1e12:.2f} TB/s)  bwd {tb*1e6:6.1f} us ({n*10/tb/1e12:.2f} TB/s)")
