"""GroupNorm(+SiLU) forward and backward through the C ABI on the denoiser's maps, timed as HIP-graph replays of 20
launches (no host launch latency in the numbers). """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd._lib import lib, check

d = torch.device("cuda:0")
BF = torch.bfloat16


def graph_time(fn, reps=20, iters=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e-3


tot_f = tot_b = 0.0
# (B, H, C, layers of that shape per train step: roughly)
for (B, H, C, cnt) in [(32, 32, 320, 13), (32, 32, 640, 2), (32, 32, 960, 1), (32, 16, 640, 13), (32, 16, 1280, 2), (32, 16, 1920, 1),
                       (32, 16, 960, 1), (32, 8, 1280, 16), (32, 8, 2560, 2), (32, 8, 1920, 1), (32, 4, 1280, 6), (32, 4, 2560, 2),
                       (4, 64, 320, 0), (4, 64, 640, 0)]:
    G = 32
    x = torch.randn(B, H, H, C, device=d).to(BF)
    dy = torch.randn(B, H, H, C, device=d).to(BF)
    ga, be = torch.randn(C, device=d), torch.randn(C, device=d)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    st = torch.empty(B, G, 2, device=d)
    dg, db = torch.empty(C, device=d), torch.empty(C, device=d)
    sc = torch.empty(int(lib.mi355_groupnorm_scratch_floats(B, H * H, C, G)), device=d)
    cs = lambda: torch.cuda.current_stream().cuda_stream
    fwd = lambda: check(lib.mi355_groupnorm_fwd(x.data_ptr(), ga.data_ptr(), be.data_ptr(), y.data_ptr(), st.data_ptr(),
                                                sc.data_ptr(), B, H * H, C, G, 1e-5, 1, cs()), "fwd")
    bwd = lambda: check(lib.mi355_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), st.data_ptr(), ga.data_ptr(), be.data_ptr(),
                                                dx.data_ptr(), dg.data_ptr(), db.data_ptr(), sc.data_ptr(), B, H * H, C, G, 1,
                                                None, cs()), "bwd")
    tf, tb = graph_time(fwd), graph_time(bwd)
    n = B * H * H * C
    tot_f += tf * cnt; tot_b += tb * cnt
    print(f"GN B{B} {H}x{H} C{C}: fwd {tf*1e6:6.1f} us ({n*4/tf/1e12:.2f} TB/s)  bwd {tb*1e6:6.1f} us ({n*6/tb/1e12:.2f} TB/s)")
print(f"weighted per train step: fwd {tot_f*1e3:.3f} ms, bwd {tot_b*1e3:.3f} ms")
