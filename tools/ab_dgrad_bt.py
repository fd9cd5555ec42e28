"""Data-gradient products from the weights as stored (k-major B, transposing LDS reads) against the same products on
transposed weight copies: the train step's shapes below the 32x32 level, timed in interleaved rounds, results compared
bit for bit.  usage: ab_dgrad_bt.py   (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd._lib import lib
if len(sys.argv) > 2:
    lib.mi355_tune_set(int(sys.argv[1]), int(sys.argv[2]))

d = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


cases = []
# conv layers: (batch, map, Cin, Cout) of the forward layer; the data gradient contracts over (tap, Cout)
for (B, H, Ci, Co) in [(32, 32, 320, 320), (32, 16, 640, 640), (32, 16, 1280, 640), (32, 16, 320, 640), (32, 8, 1280, 1280),
                       (32, 8, 2560, 1280), (32, 8, 640, 1280), (32, 4, 1280, 1280), (32, 4, 2560, 1280)]:
    dy = torch.randn(B, H, H, Co, device=d).to(BF)
    w = (torch.randn(Co, 9, Ci, device=d) * (9 * Ci) ** -0.5).to(BF)
    wt = w.flip(1).permute(2, 1, 0).contiguous()
    cases.append((f"conv3 dgrad B{B} {H}x{H} {Ci}<-{Co}", 2.0 * B * H * H * Co * 9 * Ci,
                  lambda dy=dy, wt=wt, Ci=Ci: ops.conv2d(dy, wt, Ci, 3),
                  lambda dy=dy, w=w, Ci=Ci: ops.conv2d_dgrad_bt(dy, w, Ci, 3)))
# linear layers: dx [M, K_in] = dy [M, N_out] W [N_out, K_in]
for (M, Nout, Kin) in [(8192, 640, 640), (8192, 1920, 640), (8192, 5120, 640), (8192, 640, 2560), (2048, 1280, 1280),
                       (2048, 3840, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (512, 1280, 1280), (32768, 320, 320),
                       (32768, 960, 320)]:
    dy = torch.randn(M, Nout, device=d).to(BF)
    w = (torch.randn(Nout, Kin, device=d) * Kin ** -0.5).to(BF)
    wt = w.t().contiguous()
    cases.append((f"linear dgrad M{M} {Kin}<-{Nout}", 2.0 * M * Nout * Kin,
                  lambda dy=dy, wt=wt: ops.gemm(dy, wt), lambda dy=dy, w=w: ops.gemm_bt(dy, w)))

tot = [0.0, 0.0]
for name, fl, f0, f1 in cases:
    best = [1e9, 1e9]
    for rnd in range(5):
        for i, fn in enumerate((f0, f1)):
            best[i] = min(best[i], timeit(fn))
    same = torch.equal(f0(), f1())
    tot[0] += best[0]; tot[1] += best[1]
    print(f"{name:38s} transposed copy {best[0]*1e6:7.1f} us {fl/best[0]/1e12:6.0f} TF | as stored {best[1]*1e6:7.1f} us "
          f"{fl/best[1]/1e12:6.0f} TF  ratio {best[1]/best[0]:.3f}  equal={same}", flush=True)
print(f"sum: transposed copy {tot[0]*1e6:.1f} us, as stored {tot[1]*1e6:.1f} us")
