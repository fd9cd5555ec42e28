"""Cost of the low-rank epilogue term: the same GEMM with and without it (training shapes of the LoRA step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
d = "cuda:0"


def timeit(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M, K, N, S in ((32768, 320, 320, 1), (32768, 320, 960, 3), (32768, 960, 320, 0), (8192, 640, 1920, 3), (8192, 1920, 640, 0),
                   (2048, 1280, 3840, 3), (2048, 3840, 1280, 0)):
    x = torch.randn(M, K, device=d).to(torch.bfloat16)
    w = (torch.randn(N, K, device=d) * K ** -0.5).to(torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(x, w))
    if S:       # forward: [N][4] factors, S stacked adapters
        W = 8 if 4 * S <= 8 else 16
        t16 = torch.randn(M, W, device=d).to(torch.bfloat16)
        t32 = torch.randn(M, 4 * S, device=d)
        b = torch.randn(N, 4, device=d)
        seg = N // S if S > 1 else 0
        t1 = timeit(lambda: ops.gemm(x, w, lora_t=t16, lora_b=b, lora_alpha=1.0, lora_seg=seg))
        t2 = timeit(lambda: ops.gemm(x, w, lora_t=t32, lora_b=b, lora_alpha=1.0, lora_seg=seg))
        print(f"fwd M{M} K{K} N{N} S{S}: plain {t0:6.1f} us | bf16 t {t1:6.1f} | fp32 t {t2:6.1f}")
    else:       # data gradient: [r][N] factors, r = 12
        dt16 = torch.randn(M, 16, device=d).to(torch.bfloat16)
        a = torch.randn(12, N, device=d)
        t1 = timeit(lambda: ops.gemm(x, w, lora_t=dt16, lora_b=a, lora_alpha=1.0, lora_rmajor=True))
        a4 = torch.randn(4, N, device=d)
        dt8 = torch.randn(M, 8, device=d).to(torch.bfloat16)
        t2 = timeit(lambda: ops.gemm(x, w, lora_t=dt8, lora_b=a4, lora_alpha=1.0, lora_rmajor=True))
        print(f"dx  M{M} K{K} N{N}: plain {t0:6.1f} us | r=12 {t1:6.1f} | r=4 {t2:6.1f}")
