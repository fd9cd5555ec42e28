"""Per-shape cost of the LayerNorm fold on the sampling step's shapes (run on the GPU box): for each transformer level
the products that write the residual stream (plain / emitting row statistics), the LayerNorm kernel, and the products
behind a norm (on the normalised activations / folded, plain and gated).  Interleaved timing, operands from HBM (a
256 MB buffer is cleared between launches of a pair so that nothing stays in the caches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops

d = torch.device("cuda:0")
BF = torch.bfloat16
flush = torch.empty(64 << 20, dtype=torch.float32, device=d)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3


for M, C in ((131072, 320), (32768, 640), (8192, 1280)):
    g = torch.Generator().manual_seed(C)
    a = torch.randn(M, C, generator=g).to(BF).to(d)
    res = torch.randn(M, C, generator=g).to(BF).to(d)
    w0 = (torch.randn(C, C, generator=g) * C ** -0.5).to(BF).to(d)
    b0 = torch.randn(C, generator=g).to(d)
    gamma, beta = torch.ones(C, device=d), torch.zeros(C, device=d)
    x, stat, parts = ops.gemm_rowstats(a, w0, bias=b0, residual=res)
    y = ops.layernorm_bf16(x, gamma, beta, 1e-5)
    print(f"M={M} C={C}: producer plain {timed(lambda: ops.gemm(a, w0, bias=b0, residual=res)):7.1f} us, with row sums "
          f"{timed(lambda: ops.gemm_rowstats(a, w0, bias=b0, residual=res)):7.1f} us; layernorm {timed(lambda: ops.layernorm_bf16(x, gamma, beta, 1e-5)):7.1f} us")
    for N, gated in ((3 * C, False), (C, False), (8 * C, True)):
        w1 = (torch.randn(N, C, generator=g) * C ** -0.5).to(d)
        b1 = torch.randn(N, generator=g).to(d)
        wp, cs, bo = ops.ln_fold_weight(w1, gamma, beta, b1)
        w16 = w1.to(BF)
        if gated:
            t0 = timed(lambda: ops.gemm_geglu(y, w16, b1))
            t1 = timed(lambda: ops.gemm_ln(x, wp, cs, bo, stat, parts, 1e-5, geglu=True))
        else:
            t0 = timed(lambda: ops.gemm(y, w16, bias=b1))
            t1 = timed(lambda: ops.gemm_ln(x, wp, cs, bo, stat, parts, 1e-5))
        print(f"    N={N:5d} {'gated' if gated else 'plain'}: on normalised input {t0:7.1f} us, folded {t1:7.1f} us")
