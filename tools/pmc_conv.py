"""The three equal-FLOP 3x3 convolutions of the denoiser (60.4 GFLOP each) for rocprofv3 --pmc passes: where the waves of
the 8-wave GEMM kernel spend their cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
d = "cuda:0"
BF = torch.bfloat16
for (B, H, C) in [(32, 32, 320), (32, 16, 640), (32, 8, 1280)]:
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(C, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    for _ in range(4):
        ops.conv2d(x, w, C, 3)
torch.cuda.synchronize()
