"""Yardstick, not product: the denoiser's dense GEMM shapes timed on this package's kernel and on the vendor GEMM
library behind torch.matmul, interleaved in one process, plus the implicit-GEMM convolutions against an explicit
im2col + vendor GEMM of the same contraction size (the im2col is NOT timed: it gives the vendor kernel the convolution's
FLOPs as a plain GEMM, the best case for it).  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops

d = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best


def lin_case(M, K, N, label=""):
    x = torch.randn(M, K, device=d).to(BF)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
    wt = w.t().contiguous()
    out = torch.empty(M, N, device=d, dtype=BF)
    fl = 2.0 * M * N * K
    t_mine = timeit(lambda: ops.gemm(x, w))
    t_nt = timeit(lambda: torch.matmul(x, w.t(), out=out))
    t_nn = timeit(lambda: torch.matmul(x, wt, out=out))
    t_lib = min(t_nt, t_nn)
    print(f"{label:6s} M{M:6d} K{K:6d} N{N:6d}: mine {t_mine*1e6:7.1f} us {fl/t_mine/1e12:7.1f} TF | vendor {t_lib*1e6:7.1f} us "
          f"{fl/t_lib/1e12:7.1f} TF  (NT {t_nt*1e6:6.1f}, NN {t_nn*1e6:6.1f})  mine/vendor {t_mine/t_lib:5.2f}", flush=True)


def conv_case(B, H, C, Co):
    x = torch.randn(B, H, H, C, device=d).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=d) * (9 * C) ** -0.5).to(BF)
    fl = 2.0 * B * H * H * Co * 9 * C
    t_mine = timeit(lambda: ops.conv2d(x, w, Co, 3))
    M, K = B * H * H, 9 * C
    cols = torch.randn(M, K, device=d).to(BF)
    w2 = w.reshape(Co, K)
    out = torch.empty(M, Co, device=d, dtype=BF)
    t_lib = timeit(lambda: torch.matmul(cols, w2.t(), out=out))
    print(f"conv3  B{B} {H}x{H} {C}->{Co} (M{M} K{K} N{Co}): mine {t_mine*1e6:7.1f} us {fl/t_mine/1e12:7.1f} TF | vendor GEMM on "
          f"pre-built columns {t_lib*1e6:7.1f} us {fl/t_lib/1e12:7.1f} TF  mine/vendor {t_mine/t_lib:5.2f}", flush=True)


if __name__ == "__main__":
    for M, K, N in [(32768, 320, 320), (32768, 320, 960), (32768, 320, 2560), (32768, 1280, 320), (32768, 2560, 320),
                    (8192, 640, 640), (8192, 640, 1920), (8192, 640, 5120), (8192, 2560, 640), (8192, 5120, 640),
                    (2048, 1280, 1280), (2048, 1280, 3840), (2048, 1280, 10240), (2048, 5120, 1280), (2048, 10240, 1280),
                    (512, 1280, 1280), (2464, 768, 640), (2464, 768, 2560), (8192, 8192, 8192)]:
        lin_case(M, K, N)
    for B, H, C, Co in [(32, 32, 320, 320), (32, 32, 640, 320), (32, 16, 640, 640), (32, 16, 1280, 640), (32, 8, 1280, 1280),
                        (32, 8, 2560, 1280), (32, 4, 1280, 1280)]:
        conv_case(B, H, C, Co)
