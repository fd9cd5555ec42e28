"""One representative launch set for rocprofv3 --pmc passes (run each counter group in its own process):
the 3x3 convolution 960->320 at 32x32, batch 32 (forward, data gradient operand shape, weight gradient),
GroupNorm+SiLU on the same map and the fused AdamW on 64 Mi parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd.ops import lib, check, _stream, GroupNormFn

d = torch.device("cuda:0")
BF = torch.bfloat16
B, H, C, Co = 32, 32, 960, 320
x = torch.randn(B, H, H, C, device=d).to(BF)
w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
dy = torch.randn(B, H, H, Co, device=d).to(BF)
gamma, beta = torch.ones(C, device=d), torch.zeros(C, device=d)
n = 64 << 20
p, g, m, v = (torch.randn(n, device=d) for _ in range(4))
v.abs_()
p16 = torch.empty(n, dtype=BF, device=d)
hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.001, 1.0, 1.0], device=d)
state = torch.ones(2, device=d)
for _ in range(3):
    ops.conv2d(x, w, Co, 3)
    ops.wgrad(dy, x, C, Co, 3, 1, False, geom=(B, H, H, H, H))
    GroupNormFn.apply(x, gamma, beta, 32, 1e-5, True)
    check(lib.mi355_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p16.data_ptr(), hyper.data_ptr(),
                               state.data_ptr(), n, _stream()))
torch.cuda.synchronize()
print("algorithmic: conv fwd flops", 2.0 * B * H * H * Co * 9 * C, "| conv operand bytes", x.numel() * 2 + w.numel() * 2 + dy.numel() * 2,
      "| gn bytes (3 passes)", 3 * x.numel() * 2, "| adamw bytes", 30 * n)
