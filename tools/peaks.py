"""Sustained MFMA rate (register-only v_mfma_f32_16x16x32_bf16 loops) and streaming-copy bandwidth of this GPU, the
measured counterparts of the 2.5 PFLOP/s and 8 TB/s figures the rooflines are priced against."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd._lib import lib, check

d = torch.device("cuda:0")
s = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


out = {}
sink = torch.zeros(4, device=d)
for pattern, name in ((0, "zeros"), (1, "random")):
    for waves_per_simd in (1, 2):
        blocks, iters = 256 * waves_per_simd, 200000        # one workgroup = four waves = one per SIMD of a CU
        flops = blocks * 4 * iters * 8 * 2 * 16 * 16 * 32
        t = timed(lambda: check(lib.mi355_peak_mfma_bf16(sink.data_ptr(), blocks, iters, pattern, s()), "mfma"), 30)
        out[f"mfma_bf16_{name}_{waves_per_simd}wave_TFLOPs"] = round(flops / t / 1e12, 1)
for mb in (256, 1024, 4096):
    n = mb << 20
    a, b = torch.empty(n, dtype=torch.uint8, device=d), torch.empty(n, dtype=torch.uint8, device=d)
    a.fill_(1)
    t = timed(lambda: check(lib.mi355_peak_copy(a.data_ptr(), b.data_ptr(), n, s()), "copy"), 20)
    out[f"copy_{mb}MiB_TBs"] = round(2 * n / t / 1e12, 2)
    del a, b
print(json.dumps(out))
