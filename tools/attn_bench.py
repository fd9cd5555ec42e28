import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd.ops import AttentionFn
d = "cuda:0"
def run(B, H, N, D, T=None):
    C = H * D
    if T is None:
        a = torch.randn(B, N, 3 * C, device=d).to(torch.bfloat16).requires_grad_(True); b = None; Nk = N
    else:
        a = torch.randn(B, N, C, device=d).to(torch.bfloat16).requires_grad_(True)
        b = torch.randn(B, T, 2 * C, device=d).to(torch.bfloat16).requires_grad_(True); Nk = T
    do = torch.randn(B, N, C, device=d).to(torch.bfloat16)
    def f():
        return AttentionFn.apply(a, b, None, H)
    for _ in range(3): o = f(); o.backward(do)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    it = 10
    e[0].record()
    for _ in range(it): o = f()
    e[1].record()
    for _ in range(it): o.backward(do, retain_graph=True)
    e[2].record(); torch.cuda.synchronize()
    tf, tb = e[0].elapsed_time(e[1]) / it * 1e-3, e[1].elapsed_time(e[2]) / it * 1e-3
    fl = 4.0 * B * H * N * Nk * D
    print(f"B{B} H{H} N{N} Nk{Nk} D{D}: fwd {tf*1e6:8.1f} us {fl/tf/1e12:6.1f} TF | bwd {tb*1e6:8.1f} us {2.5*fl/tb/1e12:6.1f} TF")
def run_prescaled(B, H, N, D):
    """forward only, on queries that carry scale * log2(e) (the no-gradient denoiser call's kernel)"""
    from image_generator_amd import ops
    C = H * D
    q, k, v = (torch.randn(B, N, C, device=d).to(torch.bfloat16) for _ in range(3))
    q = (q.float() * (D ** -0.5 * ops.ATTN_PRESCALE)).to(torch.bfloat16)
    for _ in range(3): ops.attention_fwd(q, k, v, H, prescaled=True)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10): ops.attention_fwd(q, k, v, H, prescaled=True)
    e[1].record(); torch.cuda.synchronize()
    t = e[0].elapsed_time(e[1]) / 10 * 1e-3
    print(f"prescaled B{B} H{H} N{N} D{D}: fwd {t*1e6:8.1f} us {4.0*B*H*N*N*D/t/1e12:6.1f} TF")
if len(sys.argv) > 1 and sys.argv[1] == "--prescaled":
    run_prescaled(*[int(v) for v in sys.argv[2:]])
elif len(sys.argv) > 1:      # one shape: attn_bench.py B H N D [T]  (under rocprofv3 --stats the per-kernel averages are then this shape's)
    run(*[int(v) for v in sys.argv[1:]])
else:
    run(32, 8, 1024, 40); run(32, 8, 256, 80); run(32, 8, 64, 160); run(32, 8, 1024, 40, 77); run(32, 8, 256, 80, 77)
    run(32, 8, 4096, 40); run(32, 8, 1024, 80); run(32, 8, 256, 160)
