"""One short-K GEMM for rocprofv3 --pmc passes: M131072 N2560 at K from argv (default 320)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
d = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 320
M, N = 131072, 2560
x = torch.randn(M, K, device=d).to(torch.bfloat16)
w = (torch.randn(N, K, device=d) * K ** -0.5).to(torch.bfloat16)
y = torch.empty(M, N, device=d, dtype=torch.bfloat16)
for _ in range(4):
    ops.gemm(x, w, out=y)
torch.cuda.synchronize()
