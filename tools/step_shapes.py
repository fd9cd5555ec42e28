"""Record every GEMM / conv launch of one real training step of the headline configuration, then time each
unique call signature stand-alone and print where the contraction time goes (run on the GPU box)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_generator_amd import ops
from image_generator_amd.trainer import DenoiserTrainer
from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig

d = torch.device("cuda:0")
BF = torch.bfloat16
calls = collections.OrderedDict()
_gemm, _conv = ops.gemm, ops.conv2d


def rec_gemm(a, w, bias=None, rowbias=None, rowbias_div=1, residual=None, out_fp32=False, alpha=1.0, **kw):
    M, K = a.shape
    key = ("gemm", M, K, w.shape[0], bias is not None, rowbias is not None, residual is not None, out_fp32, a.stride(0))
    calls[key] = calls.get(key, 0) + 1
    return _gemm(a, w, bias=bias, rowbias=rowbias, rowbias_div=rowbias_div, residual=residual, out_fp32=out_fp32,
                 alpha=alpha, **kw)


def rec_conv(x, w, cout, ksize, stride=1, ups=False, dil=False, out_hw=None, bias=None, rowbias=None, residual=None,
             out_fp32=False):
    B, H, W, C = x.shape
    key = ("conv", B, H, C, cout, ksize, stride, ups, dil, out_hw, bias is not None, rowbias is not None,
           residual is not None, out_fp32)
    calls[key] = calls.get(key, 0) + 1
    return _conv(x, w, cout, ksize, stride, ups, dil, out_hw, bias, rowbias, residual, out_fp32)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    torch.manual_seed(0)
    cfg = DenoiserConfig.sd_v1()
    with torch.device(d):
        unet = UNet2DConditionModel(cfg)
    B = 32
    sampler = len(sys.argv) > 1 and sys.argv[1] == "sampler"      # forward only at 512x512 (latent 64x64)
    side = 64 if sampler else 32
    lat, noise = torch.randn(B, 4, side, side, device=d), torch.randn(B, 4, side, side, device=d)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=d).to(BF)
    ts = torch.randint(0, 1000, (B,), device=d)
    if sampler:
        with torch.no_grad():
            unet(lat, ts, encoder_hidden_states=ctx)
            ops.gemm, ops.conv2d = rec_gemm, rec_conv
            unet(lat, ts, encoder_hidden_states=ctx)
    else:
        tr = DenoiserTrainer(unet, lr=1e-4)
        tr.train_step(lat, noise, ts, ctx)
        ops.gemm, ops.conv2d = rec_gemm, rec_conv
        tr.train_step(lat, noise, ts, ctx)
        del tr
    ops.gemm, ops.conv2d = _gemm, _conv
    torch.cuda.synchronize()
    del unet
    torch.cuda.empty_cache()

    rows = []
    for key, n in calls.items():
        if key[0] == "gemm":
            _, M, K, N, hb, hrb, hres, f32, lda = key
            a = torch.randn(M, max(lda, K), device=d).to(BF)[:, :K]
            w = (torch.randn(N, K, device=d) * K ** -0.5).to(BF)
            bias = torch.zeros(N, device=d) if hb else None
            res = torch.randn(M, N, device=d).to(BF) if hres else None
            if hrb:
                continue
            t = timeit(lambda: _gemm(a, w, bias=bias, residual=res, out_fp32=f32))
            fl = 2.0 * M * N * K
            name = f"gemm M{M} K{K} N{N}{' +b' if hb else ''}{' +res' if hres else ''}{' f32' if f32 else ''}{' strided' if lda != K else ''}"
        else:
            _, Bc, H, C, co, k, s, ups, dil, ohw, hb, hrb, hres, f32 = key
            x = torch.randn(Bc, H, H, C, device=d).to(BF)
            w = (torch.randn(co, k, k, C, device=d) * (k * k * C) ** -0.5).to(BF)
            bias = torch.zeros(co, device=d) if hb else None
            y = _conv(x, w, co, k, s, ups, dil, ohw, bias)
            Ho = y.shape[1]
            rb = torch.zeros(Bc, co, device=d) if hrb else None
            res = torch.randn_like(y) if (hres and not f32) else None
            t = timeit(lambda: _conv(x, w, co, k, s, ups, dil, ohw, bias, rb, res, f32))
            fl = 2.0 * Bc * Ho * Ho * co * k * k * C
            name = f"conv{k} B{Bc} {H}->{Ho} {C}->{co}{' s2' if s == 2 else ''}{' ups' if ups else ''}{' dil' if dil else ''}{' +rb' if hrb else ''}{' +res' if hres else ''}"
        rows.append((t * n, n, t, fl, name))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    totfl = sum(r[1] * r[3] for r in rows)
    print(f"total {tot*1e3:.2f} ms, {totfl/1e12:.2f} TFLOP, {totfl/tot/1e12:.0f} TF average")
    for tt, n, t, fl, name in rows:
        print(f"{name:58s} x{n:3d} {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF  total {tt*1e3:6.3f} ms")


if __name__ == "__main__":
    main()
