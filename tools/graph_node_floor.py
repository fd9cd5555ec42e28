"""Cost of one dependent kernel node inside a replayed HIP graph: 1 000 launches of the library's clearing kernel on the
same buffer, captured once, replayed ten times (run on the GPU box)."""
import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from image_generator_amd._lib import lib
d = torch.device("cuda:0")
buf = torch.zeros(1 << 20, device=d)
big = torch.zeros(64 << 20, device=d)
def run(n, numel, t):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            lib.mi355_zero_f32(t.data_ptr(), numel, s.cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            cs = torch.cuda.current_stream().cuda_stream
            for _ in range(n):
                lib.mi355_zero_f32(t.data_ptr(), numel, cs)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 / n * 1e6
for numel in (256, 1 << 16, 1 << 20, 16 << 20, 64 << 20):
    t = big if numel > buf.numel() else buf
    print(f"zero_f32 of {numel*4/1e6:8.3f} MB: {run(1000, numel, t):6.2f} us per graph node (1000 dependent launches)")
