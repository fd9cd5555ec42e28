"""Micro-benchmark of the contraction kernels on the denoiser's own shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops

d = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def conv_case(B, H, C, Co, k=3):
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, k, k, C, device=d) * (k * k * C) ** -0.5).to(BF)
    dy = torch.randn(B, H, H, Co, device=d).to(BF)
    fl = 2.0 * B * H * H * Co * k * k * C
    t = timeit(lambda: ops.conv2d(x, w, Co, k))
    tw = timeit(lambda: ops.wgrad(dy, x, C, Co, k, 1, False, geom=(B, H, H, H, H)))
    print(f"conv{k}x{k} B{B} {H}x{H} {C}->{Co}: fwd {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF | wgrad {tw*1e6:8.1f} us {fl/tw/1e12:7.1f} TF")


def lin_case(M, K, N):
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    dy = torch.randn(M, N, device=d).to(BF)
    fl = 2.0 * M * N * K
    t = timeit(lambda: ops.gemm(x, w))
    tw = timeit(lambda: ops.wgrad(dy, x, K, N))
    print(f"linear M{M} K{K} N{N}: fwd {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF | wgrad {tw*1e6:8.1f} us {fl/tw/1e12:7.1f} TF")


if __name__ == "__main__":
    B = 32
    conv_case(B, 32, 320, 320)
    conv_case(B, 32, 640, 320)
    conv_case(B, 32, 960, 320)
    conv_case(B, 16, 640, 640)
    conv_case(B, 16, 1280, 640)
    conv_case(B, 16, 1920, 640)
    conv_case(B, 8, 1280, 1280)
    conv_case(B, 8, 2560, 1280)
    conv_case(B, 4, 1280, 1280)
    conv_case(B, 4, 2560, 1280)
    lin_case(32768, 320, 2560)
    lin_case(32768, 1280, 320)
    lin_case(32768, 320, 960)
    lin_case(32768, 320, 320)
    lin_case(8192, 640, 5120)
    lin_case(8192, 2560, 640)
    lin_case(2048, 1280, 10240)
    lin_case(2048, 5120, 1280)
    lin_case(2464, 768, 640)
    lin_case(8192, 8192, 8192)
