"""Per-kernel parity: each HIP kernel (called through the C ABI) against plain fp32 torch math on
the same seeded inputs.  Inputs are rounded to bf16 first so both sides see identical operands;
tolerances are stated per test (bf16 outputs carry 2^-8 relative rounding)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def close(got, ref, rtol, atol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g}, ref max {float(ref.abs().max()):.4g}"


def rel_l2(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-12))


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (300, 640, 768), (128, 64, 128), (1000, 128, 72), (64, 1280, 5120),
                                   (512, 1280, 11520), (77, 2560, 1280)])
def test_gemm_plain(M, N, K):
    from image_generator_amd import ops
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    out = ops.gemm(a.to(dev()), w.to(dev()))
    ref = a.float() @ w.float().t()
    close(out, ref, 2 ** -7, 2e-2, f"gemm {M}x{N}x{K}")
    assert rel_l2(out, ref) < 4e-3


def test_gemm_epilogues():
    from image_generator_amd import ops
    M, N, K, r = 512, 320, 640, 4
    a, w = rnd(M, K, seed=3), rnd(N, K, scale=K ** -0.5, seed=4)
    bias = torch.randn(N)
    rowbias = torch.randn(M // 128, N)
    res = rnd(M, N, seed=5)
    lt = torch.randn(M, r)
    lb = torch.randn(N, r) * 0.1
    d = dev()
    out = ops.gemm(a.to(d), w.to(d), bias=bias.to(d), rowbias=rowbias.to(d), rowbias_div=128, residual=res.to(d),
                   lora_t=lt.to(d), lora_b=lb.to(d), lora_alpha=0.5)
    ref = a.float() @ w.float().t() + bias + rowbias.repeat_interleave(128, 0) + res.float() + 0.5 * lt @ lb.t()
    close(out, ref, 2 ** -7, 2e-2, "gemm epilogue")
    out32 = ops.gemm(a.to(d), w.to(d), bias=bias.to(d), out_fp32=True)
    close(out32, a.float() @ w.float().t() + bias, 1e-4, 1e-3, "gemm fp32 out")


def _conv_ref(x_nhwc, w_okki, bias, stride, ups):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    w = w_okki.float().permute(0, 3, 1, 2)
    k = w.shape[-1]
    return F.conv2d(x, w, bias, stride=stride, padding=k // 2).permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,W,C,Co,k,stride,ups", [
    (2, 16, 16, 64, 128, 3, 1, False), (2, 16, 16, 320, 320, 3, 1, False), (2, 16, 16, 128, 128, 3, 2, False),
    (2, 8, 8, 64, 64, 3, 1, True), (3, 8, 8, 8, 320, 3, 1, False), (2, 8, 8, 320, 8, 3, 1, False),
    (2, 16, 16, 640, 320, 1, 1, False), (1, 4, 4, 2560, 1280, 3, 1, False), (2, 7, 9, 72, 64, 3, 2, False)])
def test_conv2d(B, H, W, C, Co, k, stride, ups):
    from image_generator_amd import ops
    x = rnd(B, H, W, C, seed=6)
    w = rnd(Co, k, k, C, scale=(k * k * C) ** -0.5, seed=7)
    bias = torch.randn(Co) * 0.1
    d = dev()
    out = ops.conv2d(x.to(d), w.to(d), Co, k, stride=stride, ups=ups, bias=bias.to(d))
    ref = _conv_ref(x, w, bias, stride, ups)
    assert tuple(out.shape) == tuple(ref.shape)
    close(out, ref, 2 ** -7, 2e-2, "conv2d")


def test_conv2d_rowbias_residual():
    from image_generator_amd import ops
    B, H, W, C, Co = 3, 8, 8, 64, 160
    x, w = rnd(B, H, W, C, seed=8), rnd(Co, 3, 3, C, scale=(9 * C) ** -0.5, seed=9)
    rb = torch.randn(B, Co)
    res = rnd(B, H, W, Co, seed=10)
    d = dev()
    out = ops.conv2d(x.to(d), w.to(d), Co, 3, rowbias=rb.to(d), residual=res.to(d))
    ref = _conv_ref(x, w, None, 1, False) + rb[:, None, None, :] + res.float()
    close(out, ref, 2 ** -7, 2e-2, "conv2d rowbias+residual")


@pytest.mark.parametrize("B,H,W,C,Co,k,stride,ups", [
    (2, 16, 16, 64, 128, 3, 1, False), (2, 16, 16, 128, 64, 3, 2, False), (2, 8, 8, 64, 64, 3, 1, True),
    (4, 8, 8, 320, 320, 3, 1, False), (2, 8, 8, 128, 64, 1, 1, False), (2, 9, 7, 72, 64, 3, 2, False)])
def test_conv_backward(B, H, W, C, Co, k, stride, ups):
    """Data and weight gradients of the convolution against torch autograd."""
    from image_generator_amd import ops
    from image_generator_amd.ops import lib, check, _stream
    x = rnd(B, H, W, C, seed=11)
    w = rnd(Co, k, k, C, scale=(k * k * C) ** -0.5, seed=12)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    ref = _conv_ref(xr, wr, None, stride, ups)
    dy = rnd(*ref.shape, seed=13)
    ref.backward(dy.float())
    d = dev()
    wt = torch.empty((C, k, k, Co), dtype=BF, device=d)
    wd = w.to(d)
    check(lib.mi355_conv_weight_transpose(wd.data_ptr(), wt.data_ptr(), Co, k * k, C, _stream()))
    hv, wv = (2 * H, 2 * W) if ups else (H, W)
    dxv = ops.conv2d(dy.to(d), wt, C, k, stride=1, dil=(stride == 2), out_hw=(hv, wv))
    if ups:
        dx = torch.empty((B, H, W, C), dtype=BF, device=d)
        check(lib.mi355_sumpool2x2_bf16(dxv.data_ptr(), dx.data_ptr(), B, H, W, C, _stream()))
    else:
        dx = dxv
    assert rel_l2(dx, xr.grad) < 8e-3, rel_l2(dx, xr.grad)
    Ho, Wo = ref.shape[1], ref.shape[2]
    dbf = torch.full((Co,), -3.0, device=d)
    dw = ops.wgrad(dy.to(d), x.to(d), C, Co, k, stride, ups, geom=(B, H, W, Ho, Wo), dbias=dbf).view(Co, k, k, C)
    assert rel_l2(dw, wr.grad) < 2e-3, rel_l2(dw, wr.grad)
    close(dbf, dy.float().sum((0, 1, 2)), 1e-3, 1e-2, "conv bias gradient fused into wgrad")
    close(dw, wr.grad, 1e-2, 1e-2 * float(wr.grad.abs().max()), "wgrad conv")


@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (777, 640, 768), (32, 1280, 1280), (8192, 2560, 320)])
def test_linear_backward(M, N, K):
    from image_generator_amd import ops
    from image_generator_amd.ops import lib, check, _stream
    x, w, dy = rnd(M, K, seed=14), rnd(N, K, scale=K ** -0.5, seed=15), rnd(M, N, seed=16)
    d = dev()
    wt = torch.empty((K, N), dtype=BF, device=d)
    wd = w.to(d)
    check(lib.mi355_transpose_bf16(wd.data_ptr(), wt.data_ptr(), 1, N, K, _stream()))
    assert torch.equal(wt.cpu(), w.t().contiguous())
    dx = ops.gemm(dy.to(d), wt)
    assert rel_l2(dx, dy.float() @ w.float()) < 5e-3
    dbf = torch.full((N,), 7.0, device=d)          # poisoned: the call must overwrite it
    dw = ops.wgrad(dy.to(d), x.to(d), K, N, dbias=dbf).view(N, K)
    ref = dy.float().t() @ x.float()
    assert rel_l2(dw, ref) < 1e-3, rel_l2(dw, ref)
    db = ops.colsum(dy.to(d))[0]
    close(db, dy.float().sum(0), 1e-3, 1e-2, "colsum")
    close(dbf, dy.float().sum(0), 1e-3, 1e-2, "bias gradient fused into wgrad")
    if M % 32 == 0:
        seg = ops.colsum(dy.to(d), seg=M // 32)
        close(seg, dy.float().view(32, M // 32, N).sum(1), 1e-3, 1e-2, "colsum seg")


@pytest.mark.parametrize("B,HW,C,G,silu", [(2, 256, 320, 32, True), (3, 64, 640, 32, False), (2, 16, 2560, 32, True),
                                           (2, 64, 64, 8, True), (1, 1024, 960, 32, True), (2, 49, 1920, 32, False),
                                           (2, 64, 1280, 32, True), (4, 16, 1280, 32, False), (2, 64, 2560, 32, True), (2, 256, 1280, 32, True)])
def test_groupnorm(B, HW, C, G, silu):
    from image_generator_amd.ops import GroupNormFn
    h = int(math.isqrt(HW))
    x = (rnd(B, h, HW // h, C, seed=17).float() * 1.5 + 0.3).to(BF)
    gamma, beta = torch.randn(C) * 0.2 + 1.0, torch.randn(C) * 0.1
    dy = rnd(B, h, HW // h, C, seed=18)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.group_norm(xr.permute(0, 3, 1, 2), G, gr, br, 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    ref.backward(dy.float())
    d = dev()
    xg = x.to(d).requires_grad_(True)
    gg, bg = gamma.to(d).requires_grad_(True), beta.to(d).requires_grad_(True)
    y = GroupNormFn.apply(xg, gg, bg, G, 1e-5, silu)
    y.backward(dy.to(d))
    close(y, ref, 2 ** -7, 2e-2, "gn fwd")
    assert rel_l2(xg.grad, xr.grad) < 1e-2, rel_l2(xg.grad, xr.grad)
    assert rel_l2(gg.grad, gr.grad) < 2e-3 and rel_l2(bg.grad, br.grad) < 2e-3


@pytest.mark.parametrize("M,C", [(1000, 320), (64, 1280), (4096, 640), (7, 64)])
def test_layernorm(M, C):
    from image_generator_amd.ops import LayerNormFn
    x = rnd(M, C, seed=19)
    gamma, beta = torch.randn(C) * 0.2 + 1.0, torch.randn(C) * 0.1
    dy = rnd(M, C, seed=20)
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    ref.backward(dy.float())
    d = dev()
    xg, gg, bg = x.to(d).requires_grad_(True), gamma.to(d).requires_grad_(True), beta.to(d).requires_grad_(True)
    y = LayerNormFn.apply(xg, gg, bg, 1e-5)
    y.backward(dy.to(d))
    close(y, ref, 2 ** -7, 2e-2, "ln fwd")
    assert rel_l2(xg.grad, xr.grad) < 1e-2
    assert rel_l2(gg.grad, gr.grad) < 2e-3 and rel_l2(bg.grad, br.grad) < 2e-3


def _attn_ref(q, k, v, heads):
    B, Nq, C = q.shape
    D = C // heads
    qh = q.view(B, Nq, heads, D).transpose(1, 2)
    kh = k.view(B, -1, heads, D).transpose(1, 2)
    vh = v.view(B, -1, heads, D).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("B,H,N,D", [(2, 8, 256, 40), (2, 8, 64, 80), (2, 8, 16, 160), (1, 4, 300, 16), (1, 4, 100, 32),
                                     (1, 8, 1024, 40)])
def test_self_attention(B, H, N, D):
    from image_generator_amd.ops import AttentionFn
    C = H * D
    qkv = rnd(B, N, 3 * C, seed=21)
    do = rnd(B, N, C, seed=22)
    r = qkv.float().requires_grad_(True)
    ref = _attn_ref(r[..., :C], r[..., C:2 * C], r[..., 2 * C:], H)
    ref.backward(do.float())
    g = qkv.to(dev()).requires_grad_(True)
    o = AttentionFn.apply(g, None, None, H)
    o.backward(do.to(dev()))
    close(o, ref, 2 ** -6, 1e-2, "attn fwd")
    assert rel_l2(o, ref) < 6e-3
    assert rel_l2(g.grad, r.grad) < 1.5e-2, rel_l2(g.grad, r.grad)


@pytest.mark.parametrize("B,H,N,T,D", [(2, 8, 256, 77, 40), (2, 8, 64, 77, 160), (2, 4, 64, 16, 16), (1, 8, 16, 13, 80)])
def test_cross_attention(B, H, N, T, D):
    from image_generator_amd.ops import AttentionFn
    C = H * D
    q, kv, do = rnd(B, N, C, seed=23), rnd(B, T, 2 * C, seed=24), rnd(B, N, C, seed=25)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref = _attn_ref(qr, kvr[..., :C], kvr[..., C:], H)
    ref.backward(do.float())
    qg, kvg = q.to(dev()).requires_grad_(True), kv.to(dev()).requires_grad_(True)
    o = AttentionFn.apply(qg, kvg, None, H)
    o.backward(do.to(dev()))
    close(o, ref, 2 ** -6, 1e-2, "xattn fwd")
    assert rel_l2(qg.grad, qr.grad) < 1.5e-2, rel_l2(qg.grad, qr.grad)
    assert rel_l2(kvg.grad, kvr.grad) < 1.5e-2, rel_l2(kvg.grad, kvr.grad)


def test_attention_softmax_spike():
    """One key dominates late in the sequence so the running max jumps between key tiles."""
    from image_generator_amd.ops import AttentionFn
    B, H, N, D = 1, 2, 256, 40
    C = H * D
    qkv = rnd(B, N, 3 * C, seed=26).float()
    qkv[0, 200, C:2 * C] = qkv[0, 5, :C] * 6.0     # key 200 aligned with query 5
    qkv = qkv.to(BF)
    r = qkv.float()
    ref = _attn_ref(r[..., :C], r[..., C:2 * C], r[..., 2 * C:], H)
    o = AttentionFn.apply(qkv.to(dev()), None, None, H)
    close(o, ref, 2 ** -6, 1e-2, "attn spike")


@pytest.mark.parametrize("B,H,N,T,D", [(1, 2, 4096, 4096, 40), (1, 4, 2304, 2304, 40), (2, 8, 1024, 1024, 40), (2, 4, 300, 77, 40), (1, 3, 515, 700, 56), (2, 2, 200, 130, 24),
                                       (1, 4, 256, 256, 80), (1, 2, 64, 77, 160), (1, 2, 100, 100, 32)])
def test_attention_on_prescaled_queries(B, H, N, T, D):
    """mi355_attention_fwd_prescaled (queries that carry scale * log2(e); no running maximum in the key loop for heads
    with a padding column, the plain kernel for the others) against fp32 softmax(q k^T / sqrt(d)) v on the unscaled
    queries and against the plain entry: same tolerance as the plain kernel, log-sum-exp rows equal to 2e-2 (log2 units;
    the prescaled query is rounded to bf16 once more)."""
    from image_generator_amd import ops
    C = H * D
    q, k, v = rnd(B, N, C, seed=31), rnd(B, T, C, seed=32), rnd(B, T, C, seed=33)
    ref = _attn_ref(q.float(), k.float(), v.float(), H)
    qs = (q.float() * (D ** -0.5 * ops.ATTN_PRESCALE)).to(BF)
    o0, lse0 = ops.attention_fwd(q.to(dev()), k.to(dev()), v.to(dev()), H)
    from image_generator_amd._lib import lib
    try:
        for form in (2, 3):      # 32 queries per wave; 64 queries per wave (the long-sequence form, forced on every shape)
            lib.mi355_tune_set(0, form)
            o, lse = ops.attention_fwd(qs.to(dev()), k.to(dev()), v.to(dev()), H, prescaled=True)
            close(o, ref, 2 ** -6, 1e-2, "attn prescaled")
            assert rel_l2(o, ref) < 6e-3
            assert (lse - lse0).abs().max().item() < 2e-2
    finally:
        lib.mi355_tune_set(0, 0)


@pytest.mark.parametrize("gain,shift", [(40.0, 0.0), (1.0, -300.0), (1.0, 300.0), (25.0, -150.0)])
def test_attention_on_prescaled_queries_leaves_the_fast_range(gain, shift):
    """Scores far outside +-100 (log2 units): huge positive ones overflow the loop without a running maximum, hugely
    negative ones underflow it - the workgroup must notice from its row sums and repeat the loop with the maximum.
    shift: a common offset on every score of a row (one extra channel: q = shift, k = 1) - softmax does not see it."""
    from image_generator_amd import ops
    B, H, N, D = 1, 2, 300, 40
    C = H * D
    q, k, v = rnd(B, N, C, seed=34).float() * gain, rnd(B, N, C, seed=35).float(), rnd(B, N, C, seed=36)
    q = q.view(B, N, H, D); k = k.view(B, N, H, D)
    q[..., 0] = shift; k[..., 0] = 1.0               # (bf16-exact values)
    q, k = q.reshape(B, N, C).to(BF), k.reshape(B, N, C).to(BF)
    # the prescaled operand is what the kernel is given; the reference divides the scale out again so both see the same q
    c = D ** -0.5 * ops.ATTN_PRESCALE
    qs = q.to(BF)
    ref = _attn_ref(qs.float() / c, k.float(), v.float(), H)
    s = torch.einsum("bnhd,bmhd->bhnm", qs.float().view(B, N, H, D), k.float().view(B, N, H, D))
    assert s.abs().max().item() > 150.0               # the case really leaves the range
    lse_ref = torch.logsumexp(s * math.log(2.0), -1) / math.log(2.0)
    from image_generator_amd._lib import lib
    try:
        for form in (2, 3):      # 32 and 64 queries per wave
            lib.mi355_tune_set(0, form)
            o, lse = ops.attention_fwd(qs.to(dev()), k.to(dev()), v.to(dev()), H, prescaled=True)
            assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
            close(o, ref, 2 ** -6, 2e-2, "attn prescaled, extreme scores")
            assert ((lse.cpu() - lse_ref).abs() / lse_ref.abs().clamp_min(1.0)).max().item() < 1e-2
    finally:
        lib.mi355_tune_set(0, 0)


def test_table_driven_copies():
    """mi355_copy_ranges: n (source, destination, bytes) records in one launch; ranges of very different lengths, one of
    them a single 16-byte unit, destinations untouched outside the ranges."""
    from image_generator_amd._lib import lib, check
    torch.manual_seed(41)
    src = torch.randn(200_000, device=dev())
    dst = torch.full((300_000,), -7.0, device=dev())
    recs = [(0, 1000, 4), (4, 8, 100_000), (120_000, 150_004, 33_332), (199_996, 299_996, 4)]     # element offsets, lengths
    table = torch.tensor([[src.data_ptr() + 4 * a, dst.data_ptr() + 4 * b, 4 * n] for a, b, n in recs], dtype=torch.int64,
                         device=dev())
    for blocks in (1, 64):
        dst.fill_(-7.0)
        check(lib.mi355_copy_ranges(table.data_ptr(), len(recs), blocks, torch.cuda.current_stream().cuda_stream), "copy_ranges")
        want = torch.full_like(dst, -7.0)
        for a, b, n in recs:
            want[b:b + n] = src[a:a + n]
        assert torch.equal(dst, want)


def test_geglu_and_silu():
    from image_generator_amd.ops import GegluFn, SiluF32ToBf16Fn
    x, dy = rnd(300, 2 * 640, seed=27), rnd(300, 640, seed=28)
    xr = x.float().requires_grad_(True)
    h, g = xr.chunk(2, -1)
    ref = h * F.gelu(g)
    ref.backward(dy.float())
    xg = x.to(dev()).requires_grad_(True)
    y = GegluFn.apply(xg)
    y.backward(dy.to(dev()))
    close(y, ref, 2 ** -7, 1e-2, "geglu fwd")
    assert rel_l2(xg.grad, xr.grad) < 6e-3
    t = torch.randn(32, 1280)
    tr = t.clone().requires_grad_(True)
    F.silu(tr).backward(torch.ones_like(t))
    tg = t.to(dev()).requires_grad_(True)
    s = SiluF32ToBf16Fn.apply(tg)
    s.backward(torch.ones_like(s))
    close(s, F.silu(t), 2 ** -7, 1e-2, "silu")
    close(tg.grad, tr.grad, 1e-4, 1e-5, "silu bwd")


def test_timestep_embedding_and_layout():
    from image_generator_amd import ops
    t = torch.tensor([0.0, 1.0, 37.0, 999.0])
    e = ops.timestep_embedding(t.to(dev()), 320)
    half = 160
    f = torch.exp(-math.log(10000.0) * torch.arange(half) / half)
    ref = torch.cat([torch.cos(t[:, None] * f), torch.sin(t[:, None] * f)], -1)
    close(e, ref, 2 ** -7, 2e-2, "timestep embedding")   # bf16 output; fp32 sin/cos of large arguments
    x = torch.randn(2, 4, 5, 6)
    y = ops.nchw_to_nhwc_bf16(x.to(dev()), 8)
    assert torch.equal(y[..., :4].cpu(), x.permute(0, 2, 3, 1).to(BF)) and float(y[..., 4:].abs().max()) == 0.0
    z = ops.ToNCHWFn.apply(y.float(), 4)
    assert torch.equal(z.cpu(), x.to(BF).float())


def test_mse_and_schedule_ops():
    from image_generator_amd import ops
    from image_generator_amd.ops import lib, check, _stream
    from oracle import ops_ref
    d = dev()
    p, t = torch.randn(4, 4, 16, 16), torch.randn(4, 4, 16, 16)
    pg = p.to(d).requires_grad_(True)
    loss = ops.mse_loss(pg, t.to(d))
    loss.backward()
    pr = p.clone().requires_grad_(True)
    lr = ops_ref.mse_loss(pr, t)
    lr.backward()
    close(loss, lr, 1e-5, 1e-6, "mse loss")
    close(pg.grad, pr.grad, 1e-5, 1e-8, "mse grad")
    acp = ops_ref.sd_alphas_cumprod()
    ts = torch.tensor([0, 10, 500, 999])
    out = torch.empty_like(p, device=d)
    pd, td, tsd, acpd = p.to(d), t.to(d), ts.to(d), acp.to(d)     # keep every operand alive across the launch
    check(lib.mi355_ddpm_add_noise(pd.data_ptr(), td.data_ptr(), tsd.data_ptr(), acpd.data_ptr(),
                                   out.data_ptr(), 4, p[0].numel(), _stream()))
    close(out, ops_ref.ddpm_add_noise(p, t, ts, acp), 1e-5, 1e-6, "add_noise")
    eps = torch.randn(2, 4, 4, 16, 16)
    x = torch.randn(4, 4, 16, 16)
    tt, g = 481, 7.5
    coef = torch.tensor([float(acp[tt]), float(acp[tt - 20]), g])
    o = torch.empty_like(x, device=d)
    epsd, xd, coefd = eps.to(d), x.to(d), coef.to(d)
    check(lib.mi355_ddim_step_cfg(epsd.data_ptr(), xd.data_ptr(), o.data_ptr(), coefd.data_ptr(),
                                  x.numel(), 1, _stream()))
    e = ops_ref.cfg_combine(eps[0], eps[1], g)
    close(o, ops_ref.ddim_step(e, tt, x, acp, 50), 2e-5, 2e-5, "ddim step")


def test_sumsq_ranges_and_wgrad_job_slots():
    """mi355_sumsq_ranges: squares over aligned, unaligned, ragged and empty-tail ranges plus a range of ready partial
    sums (negative count), against float64.  A single unsplit weight-gradient job with `sumsq_slots` leaves the sum of
    squares of exactly the gradient it stored in the slots; a pixel-split job leaves them untouched."""
    import ctypes
    from image_generator_amd.ops import lib, check, _stream, WgradQueue
    d = dev()
    g = torch.Generator().manual_seed(3)
    buf = torch.randn(300_000, generator=g).to(d)
    parts = torch.rand(64, generator=g).to(d)
    spans = [(0, 100_000), (100_001, 37), (100_040, 3), (150_003, 149_997)]
    rows = [(buf.data_ptr() + 4 * o, n) for o, n in spans] + [(parts.data_ptr(), -64)]
    table = torch.tensor(rows, dtype=torch.int64).to(d)
    out = torch.full((1,), 2.0, device=d)
    check(lib.mi355_sumsq_ranges(table.data_ptr(), len(rows), out.data_ptr(), _stream()))
    want = 2.0 + sum(float(buf[o:o + n].double().pow(2).sum()) for o, n in spans) + float(parts.double().sum())
    assert abs(float(out) - want) <= 1e-5 * want, (float(out), want)

    for B, H, split in ((2, 8, False), (16, 64, True)):
        C, Co = 32, 48
        x = torch.randn(B, H, H, C, generator=g).to(d).to(BF)
        dy = torch.randn(B, H, H, Co, generator=g).to(d).to(BF)
        dw = torch.zeros(Co, 9, C, device=d)
        slots = torch.zeros(int(lib.mi355_wgrad_sumsq_slots()), device=d)
        q = WgradQueue(d)
        q.zero_split_outputs = True
        q.track_unsplit(dw.data_ptr(), dw.data_ptr() + 4 * dw.numel(), slots=slots)
        q.push(dy, x, dw, None, B, H, H, C, Co, 3, 1, False, H, H)
        q.join()
        torch.cuda.synchronize()
        assert (len(q.covered) == 0) == split, (B, H, q.covered)
        want = 0.0 if split else float(dw.double().pow(2).sum())
        assert float(dw.abs().max()) > 0
        assert abs(float(slots.double().sum()) - want) <= 1e-5 * max(want, 1.0), (split, float(slots.sum()), want)
        # the same job with a bf16 destination: unsplit -> bf16 there (round to nearest even) and dw untouched;
        # split -> the bf16 buffer untouched, fp32 accumulated into dw as before
        dw2 = torch.zeros_like(dw)
        st16 = torch.full((Co, 9, C), 7.0, dtype=BF, device=d)
        q.track_unsplit(dw2.data_ptr(), dw2.data_ptr() + 4 * dw2.numel(), stage16=st16)
        q.push(dy, x, dw2, None, B, H, H, C, Co, 3, 1, False, H, H)
        q.join()
        torch.cuda.synchronize()
        if split:
            assert float((st16.float() - 7.0).abs().max()) == 0 and q.covered_map == {}
            close(dw2, dw, 1e-4, 1e-4 * float(dw.abs().max()), "split wgrad with bf16 destination")
        else:
            assert q.covered_map == {dw2.data_ptr(): dw2.numel()}
            assert float(dw2.abs().max()) == 0
            assert torch.equal(st16, dw.to(BF))


def test_adamw_matches_torch():
    from image_generator_amd.ops import lib, check, _stream
    d = dev()
    n = 4096 + 8
    p0, g = torch.randn(n), torch.randn(n) * 3
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2, weight_decay=0.01)
    p = p0.to(d)
    m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
    p16 = torch.empty(n, dtype=BF, device=d)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        torch.nn.utils.clip_grad_norm_([pt], 1.0)
        opt.step()
        gg = (g * step).to(d)
        state = torch.zeros(2, device=d)
        check(lib.mi355_sumsq_f32(gg.data_ptr(), state.data_ptr(), n, _stream()))
        hyper = torch.tensor([1e-2, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9 ** step, 1 - 0.999 ** step, 1.0, 1.0], device=d)
        check(lib.mi355_adamw_step(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), p16.data_ptr(),
                                   hyper.data_ptr(), state.data_ptr(), n, _stream()))
    close(p, pt.detach(), 1e-5, 1e-6, "adamw")
    assert torch.equal(p16.cpu(), p.cpu().to(BF))


def test_contrastive_loss_golden(golden_dir):
    """Fused InfoNCE against vectors minted from /root/reference/loss.py (fp32; tolerance 1e-4 rel)."""
    import os
    from image_generator_amd.loss import ContrastiveLoss
    z = np.load(os.path.join(golden_dir, "contrastive.npz"))
    for ci in range(4):
        t = torch.tensor(z[f"c{ci}_t"]).to(dev()).requires_grad_(True)
        i = torch.tensor(z[f"c{ci}_i"]).to(dev()).requires_grad_(True)
        m = ContrastiveLoss().to(dev())
        with torch.no_grad():
            m.log_temp.copy_(torch.tensor(z[f"c{ci}_log_temp"]))
        loss = m(t, i)
        loss.backward()
        scale = max(1.0, abs(float(z[f"c{ci}_loss"])))
        close(loss, torch.tensor(z[f"c{ci}_loss"]), 2e-4, 2e-5 * scale, f"contrastive loss c{ci}")
        for got, key in ((t.grad, "dt"), (i.grad, "di"), (m.log_temp.grad, "dlog_temp")):
            ref = torch.tensor(z[f"c{ci}_{key}"])
            close(got, ref, 2e-3, 2e-5 * max(1.0, float(ref.abs().max())), f"contrastive {key} c{ci}")


def test_lora_linear_golden(golden_dir):
    """LoRALinear forward/backward against vectors minted from /root/reference/lora.py.
    The kernel computes in bf16 (fp32 accumulate): tolerance 2^-6 relative on y, 2% L2 on grads."""
    import os
    from image_generator_amd.lora import LoRALinear
    z = np.load(os.path.join(golden_dir, "lora.npz"))
    for ci in range(2):
        w, b = torch.tensor(z[f"c{ci}_w"]), torch.tensor(z[f"c{ci}_b"])
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=b.numel() > 0)
        with torch.no_grad():
            lin.weight.copy_(w)
            if b.numel():
                lin.bias.copy_(b)
        m = LoRALinear(lin, r=4, alpha=float(z[f"c{ci}_alpha"])).to(dev())
        with torch.no_grad():
            m.lora_A.weight.copy_(torch.tensor(z[f"c{ci}_a"]))
            m.lora_B.weight.copy_(torch.tensor(z[f"c{ci}_bb"]))
        x = torch.tensor(z[f"c{ci}_x"]).to(dev()).requires_grad_(True)
        y = m(x)
        y.backward(torch.tensor(z[f"c{ci}_go"]).to(dev()))
        ref_y = torch.tensor(z[f"c{ci}_y"])
        close(y, ref_y, 2 ** -6, 3e-2, f"lora y c{ci}")
        assert rel_l2(x.grad, torch.tensor(z[f"c{ci}_dx"])) < 2e-2
        assert rel_l2(m.lora_A.weight.grad, torch.tensor(z[f"c{ci}_da"])) < 2e-2
        assert rel_l2(m.lora_B.weight.grad, torch.tensor(z[f"c{ci}_db"])) < 2e-2
        assert m.original.weight.grad is None


def test_gemm_geglu_fused_epilogue():
    """out = (x W_h^T + b_h) * gelu_erf(x W_g^T + b_g) in one launch (inference path of the GEGLU block) against
    fp32 torch; 2^-7 relative on the bf16 output (the fused path keeps both halves in fp32 up to the product)."""
    from image_generator_amd import ops
    for (M, K, Fh) in [(300, 128, 160), (1024, 320, 1280)]:
        x = rnd(M, K, seed=1).to(dev())
        w = rnd(2 * Fh, K, scale=K ** -0.5, seed=2).to(dev())
        b = rnd(2 * Fh, seed=3).float().to(dev())
        for bias in (b, None):
            y = ops.gemm_geglu(x, w, bias)
            pre = x.float() @ w.float().t() + (0 if bias is None else bias)
            ref = pre[:, :Fh] * F.gelu(pre[:, Fh:])
            close(y, ref, 2 ** -7, 2 ** -7 * float(ref.abs().max()), f"geglu M{M} K{K} F{Fh}")


def test_wide_tile_gemm_paths():
    """Grids of 512+ tiles of 256 x 160 take the wide-tile kernel (dense, implicit-GEMM conv, fused GEGLU): ragged M,
    bias + residual + row bias, every conv addressing mode, against fp32 torch on the GPU (CPU would take minutes)."""
    from image_generator_amd import ops
    d = dev()
    # dense: M not a multiple of 256, N = 2560, K = 320
    M, N, K = 256 * 33 - 24, 2560, 320
    a, w = rnd(M, K, seed=11).to(d), rnd(N, K, scale=K ** -0.5, seed=12).to(d)
    bias, res = torch.randn(N, generator=torch.Generator().manual_seed(13)).to(d), rnd(M, N, seed=14).to(d)
    out = ops.gemm(a, w, bias=bias, residual=res)
    ref = a.float() @ w.float().t() + bias + res.float()
    assert rel_l2(out, ref) < 4e-3
    close(out, ref, 2 ** -7, 3e-2, "wide dense")
    # convolution 3x3 (padding taps), row bias per sample, 64 channels in: B*H*W = 34*32*32 pixels x 640 outputs
    B, H, C, Co = 34, 32, 64, 640
    x, wc = rnd(B, H, H, C, seed=15).to(d), rnd(Co, 3, 3, C, scale=(9 * C) ** -0.5, seed=16).to(d)
    rb = torch.randn(B, Co, generator=torch.Generator().manual_seed(17)).to(d)
    y = ops.conv2d(x, wc, Co, 3, rowbias=rb)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wc.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) + rb[:, None, None, :]
    assert rel_l2(y, ref) < 4e-3
    # nearest-2x upsampled input and the zero-dilated (data-gradient of stride 2) input
    x2 = rnd(B, 16, 16, C, seed=18).to(d)
    y = ops.conv2d(x2, wc, Co, 3, ups=True)
    ref = F.conv2d(F.interpolate(x2.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"),
                   wc.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y, ref) < 4e-3
    y = ops.conv2d(x2, wc, Co, 3, dil=True, out_hw=(32, 32))
    xd = torch.zeros(B, C, 32, 32, device=d)
    xd[:, :, ::2, ::2] = x2.float().permute(0, 3, 1, 2)
    ref = F.conv2d(xd, wc.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y, ref) < 4e-3
    # stride 2
    x3 = rnd(B, 64, 64, C, seed=19).to(d)
    y = ops.conv2d(x3, wc, Co, 3, stride=2)
    ref = F.conv2d(x3.float().permute(0, 3, 1, 2), wc.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y, ref) < 4e-3
    # fused GEGLU with the stored pre-activation
    M, C = 256 * 36, 320
    a, w = rnd(M, C, seed=20).to(d), rnd(8 * C, C, scale=C ** -0.5, seed=21).to(d)
    bias = torch.randn(8 * C, generator=torch.Generator().manual_seed(22)).to(d)
    y, pre = ops.gemm_geglu(a, w, bias, keep_preact=True)
    pr = a.float() @ w.float().t() + bias
    assert rel_l2(pre, pr) < 4e-3
    ref = pr[:, :4 * C] * F.gelu(pr[:, 4 * C:])
    assert rel_l2(y, ref) < 6e-3


def test_norm_passthrough_and_frozen_affine():
    """Norm Functions in pass-through form: y = norm(x), x' = x; the gradient of the branch on x' is added inside
    the backward kernel (dx = norm_bwd(dy) + dpass).  Frozen affine parameters: same dx, no parameter gradients."""
    from image_generator_amd.ops import GroupNormFn, LayerNormFn
    d = dev()
    x = (rnd(2, 8, 8, 64, seed=31).float() * 1.3 + 0.2).to(BF)
    gamma, beta = torch.randn(64) * 0.2 + 1.0, torch.randn(64) * 0.1
    dy, dp = rnd(2, 8, 8, 64, seed=32), rnd(2, 8, 8, 64, seed=33)
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.silu(F.group_norm(xr.permute(0, 3, 1, 2), 8, gr, br, 1e-5)).permute(0, 2, 3, 1)
    (ref * dy.float()).sum().backward()
    want_dx = xr.grad + dp.float()
    for frozen in (False, True):
        xg = x.to(d).requires_grad_(True)
        gg, bg = gamma.to(d).requires_grad_(not frozen), beta.to(d).requires_grad_(not frozen)
        y, xp = GroupNormFn.apply(xg, gg, bg, 8, 1e-5, True, None, True)
        assert xp.data_ptr() == xg.data_ptr()
        torch.autograd.backward([y, xp], [dy.to(d), dp.to(d)])
        assert rel_l2(xg.grad, want_dx) < 1e-2
        if frozen:
            assert gg.grad is None and bg.grad is None
        else:
            assert rel_l2(gg.grad, gr.grad) < 2e-3 and rel_l2(bg.grad, br.grad) < 2e-3
    x2, dy2, dp2 = rnd(100, 320, seed=34), rnd(100, 320, seed=35), rnd(100, 320, seed=36)
    g2, b2 = torch.randn(320) * 0.2 + 1.0, torch.randn(320) * 0.1
    xr = x2.float().requires_grad_(True)
    (F.layer_norm(xr, (320,), g2, b2, 1e-5) * dy2.float()).sum().backward()
    for frozen in (False, True):
        xg = x2.to(d).requires_grad_(True)
        gg, bg = g2.to(d).requires_grad_(not frozen), b2.to(d).requires_grad_(not frozen)
        y, xp = LayerNormFn.apply(xg, gg, bg, 1e-5, None, True)
        torch.autograd.backward([y, xp], [dy2.to(d), dp2.to(d)])
        assert rel_l2(xg.grad, xr.grad + dp2.float()) < 1e-2
        assert (gg.grad is None) == frozen


def test_small_utility_entry_points():
    """zero_ranges, copy2d_batched and the bf16-gradient optimizer variants against their obvious definitions."""
    from image_generator_amd.ops import lib, check, _stream
    d = dev()
    buf = torch.ones(4096, device=d)
    table = torch.tensor([buf.data_ptr() + 4 * 16, 4 * 64, buf.data_ptr() + 4 * 1024, 4 * 2000], dtype=torch.int64, device=d)
    check(lib.mi355_zero_ranges(table.data_ptr(), 2, _stream()))
    want = torch.ones(4096)
    want[16:80] = 0
    want[1024:3024] = 0
    assert torch.equal(buf.cpu(), want)
    src = torch.arange(40 * 8, dtype=torch.float32, device=d).view(40, 8)
    dst = torch.full((40, 4), -1.0, device=d)
    src2, dst2 = torch.randn(8, 96, device=d), torch.zeros(4, 96, device=d)
    t2 = torch.tensor([src.data_ptr() + 4 * 4, dst.data_ptr(), 40, 4, 8, 4,
                       src2.data_ptr(), dst2.data_ptr(), 4, 96, 96, 96], dtype=torch.int64, device=d)
    check(lib.mi355_copy2d_f32_batched(t2.data_ptr(), 2, _stream()))
    assert torch.equal(dst, src[:, 4:8]) and torch.equal(dst2, src2[:4])
    # AdamW fed with bf16 gradients == AdamW fed with the same values in fp32
    n = 8192
    g16 = (torch.randn(n) * 2).to(BF).to(d)
    outs = []
    for use16 in (False, True):
        torch.manual_seed(5)
        p = torch.randn(n, device=d)
        m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
        p16 = torch.empty(n, dtype=BF, device=d)
        state = torch.zeros(2, device=d)
        hyper = torch.tensor([1e-2, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.001, 1.0, 0.5], device=d)
        if use16:
            check(lib.mi355_sumsq_bf16(g16.data_ptr(), state.data_ptr(), n, _stream()))
            check(lib.mi355_adamw_step_g16(p.data_ptr(), g16.data_ptr(), m.data_ptr(), v.data_ptr(), p16.data_ptr(),
                                           hyper.data_ptr(), state.data_ptr(), n, _stream()))
        else:
            g32 = g16.float()
            check(lib.mi355_sumsq_f32(g32.data_ptr(), state.data_ptr(), n, _stream()))
            check(lib.mi355_adamw_step(p.data_ptr(), g32.data_ptr(), m.data_ptr(), v.data_ptr(), p16.data_ptr(),
                                       hyper.data_ptr(), state.data_ptr(), n, _stream()))
        outs.append((p.cpu(), m.cpu(), v.cpu(), p16.float().cpu(), float(state[0])))
    assert abs(outs[0][4] - outs[1][4]) <= 1e-5 * outs[0][4]
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,K,W", [(1000, 320, 8), (77 * 5, 768, 8), (4096, 3840, 16), (33, 64, 16)])
def test_gemm_skinny(M, K, W):
    from image_generator_amd import ops
    a, w = rnd(M, K, seed=41), rnd(W, K, scale=K ** -0.5, seed=42)
    out = ops.gemm_skinny(a.to(dev()), w.to(dev()), 0.5)
    ref = 0.5 * (a.float() @ w.float().t())
    close(out, ref, 2 ** -7, 1e-2, f"skinny {M}x{W}x{K}")


def test_measurement_aids():
    """The measurement entry points do what their arithmetic says: the streaming copy copies (ragged tail included), the
    MFMA loop runs and reports a rate between a tenth of and the whole spec peak, the event-overhead probe returns a
    few microseconds, and bad arguments are refused."""
    from image_generator_amd.ops import lib, check, _stream
    d = dev()
    n = (1 << 20) + 16 * 37
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=d)
    dst = torch.zeros(n, dtype=torch.uint8, device=d)
    check(lib.mi355_peak_copy(src.data_ptr(), dst.data_ptr(), n, _stream()))
    assert torch.equal(src, dst)
    assert lib.mi355_peak_copy(src.data_ptr(), dst.data_ptr(), n - 3, _stream()) != 0
    sink = torch.zeros(4, device=d)
    assert lib.mi355_peak_mfma_bf16(sink.data_ptr(), 256, 12, 1, _stream()) != 0          # iters not a multiple of 8
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib.mi355_peak_mfma_bf16(sink.data_ptr(), 256, 8000, 1, _stream()))
    e0.record()
    check(lib.mi355_peak_mfma_bf16(sink.data_ptr(), 256, 80000, 1, _stream()))
    e1.record()
    torch.cuda.synchronize()
    tf = 256 * 4 * 80000 * 8 * 16384 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    assert 250.0 < tf < 2600.0, tf
    assert float(sink.abs().sum()) == 0.0
    cal = torch.zeros(1 << 20, device=d)
    us = lib.mi355_prof_event_overhead_us(_stream(), cal.data_ptr(), cal.numel(), 64)
    assert 0.0 <= us < 100.0, us
    assert lib.mi355_prof_event_overhead_us(_stream(), None, 8, 64) < 0


@pytest.mark.parametrize("shape", [(2, 32, 32, 640, 320), (3, 16, 16, 1280, 640), (2, 8, 8, 1280, 1280), (2, 4, 4, 1280, 1280),
                                   (1, 64, 64, 320, 320), (2, 16, 16, 640, 640)])
def test_groupnorm_on_two_sources_equals_groupnorm_of_the_concatenation(shape):
    """mi355_groupnorm_fwd_cat (hidden state | skip connection read in place) against mi355_groupnorm_fwd on the
    materialised concatenation: the same kernels compute the same sums in the same order, so the results are equal
    bit for bit - on the strip kernels (group width not a multiple of 8, or a large map) and on the small-map kernel."""
    from image_generator_amd.ops import GroupNormFn, groupnorm_cat
    d = dev()
    B, H, W, C1, C2 = shape
    g = torch.Generator().manual_seed(sum(shape))
    x1 = (torch.randn(B, H, W, C1, generator=g) * 1.5 + 0.3).to(BF).to(d)
    x2 = (torch.randn(B, H, W, C2, generator=g) * 0.7 - 0.2).to(BF).to(d)
    gamma = (torch.randn(C1 + C2, generator=g) * 0.2 + 1).to(d)
    beta = (torch.randn(C1 + C2, generator=g) * 0.1).to(d)
    for silu in (True, False):
        want = GroupNormFn.apply(torch.cat([x1, x2], -1), gamma, beta, 32, 1e-5, silu)
        got = groupnorm_cat(x1, x2, gamma, beta, 32, 1e-5, silu)
        assert torch.equal(got, want)


def test_up_path_resnet_without_the_concatenation():
    """ResnetBlock.forward_cat (inference) against the block on torch.cat([hidden, skip]): normalisation and 3x3
    convolutions see identical inputs; the 1x1 shortcut is two products whose first sum is rounded to bf16 once more."""
    from image_generator_amd.unet import ResnetBlock
    d = dev()
    torch.manual_seed(3)
    blk = ResnetBlock(640 + 320, 320, 1280, 32, 1e-5).to(d)
    x1 = rnd(2, 16, 16, 640, seed=41).to(d)
    x2 = rnd(2, 16, 16, 320, seed=42).to(d)
    temb = rnd(2, 1280, seed=43).to(d)
    import image_generator_amd.unet as U
    # a stand-alone block has no flat storage: give its layers their bf16 operand copies the way the model does
    for m in blk.modules():
        if isinstance(m, (U.HipConv2d, U.HipLinear)):
            pk = m.pack
            if isinstance(m, U.HipConv2d):
                t = pk.ksize * pk.ksize
                pk.w16 = m.weight.data.permute(0, 2, 3, 1).reshape(pk.cout, t, pk.cin).to(BF).contiguous()
            else:
                pk.w16 = m.weight.data.to(BF).contiguous()
    with torch.no_grad():
        assert blk.cat_ready(x1, x2)
        got = blk.forward_cat(x1, x2, temb)
        want = blk(torch.cat([x1, x2], -1), temb)
    assert rel_l2(got, want.float()) < 4e-3


@pytest.mark.parametrize("shape", [(2, 32, 32, 640, 320), (3, 16, 16, 1280, 640), (2, 8, 8, 1280, 1280), (2, 4, 4, 1280, 1280),
                                   (2, 8, 8, 1280, 640), (2, 16, 16, 640, 640)])
def test_groupnorm_backward_on_two_sources(shape):
    """mi355_groupnorm_bwd_cat against mi355_groupnorm_bwd on the materialised concatenation (with a bypass gradient and
    parameter gradients): the same kernels, the same sums in the same order - input gradients equal bit for bit as the two
    column blocks of the reference's, on the strip kernels and on the small-map kernel."""
    from image_generator_amd.ops import lib, check, _stream
    d = dev()
    B, H, W, C1, C2 = shape
    C, G, HW = C1 + C2, 32, H * W
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x1 = (torch.randn(B, H, W, C1, generator=g) * 1.5 + 0.3).to(BF).to(d)
    x2 = (torch.randn(B, H, W, C2, generator=g) * 0.7 - 0.2).to(BF).to(d)
    dy = torch.randn(B, H, W, C, generator=g).to(BF).to(d)
    r1 = torch.randn(B, H, W, C1, generator=g).to(BF).to(d)
    r2 = torch.randn(B, H, W, C2, generator=g).to(BF).to(d)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1).to(d)
    beta = (torch.randn(C, generator=g) * 0.1).to(d)
    xc, rc = torch.cat([x1, x2], -1), torch.cat([r1, r2], -1)
    nsc = int(lib.mi355_groupnorm_scratch_floats(B, HW, C, G))
    for silu in (1, 0):
        y = torch.empty_like(xc)
        stats = torch.empty(B, G, 2, device=d)
        sc = torch.empty(nsc, device=d)
        check(lib.mi355_groupnorm_fwd(xc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                      sc.data_ptr(), B, HW, C, G, 1e-5, silu, _stream()))
        for with_res in (True, False):
            dx = torch.empty_like(xc)
            dg, db = torch.empty(C, device=d), torch.empty(C, device=d)
            check(lib.mi355_groupnorm_bwd(xc.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), sc.data_ptr(), B, HW, C, G, silu,
                                          rc.data_ptr() if with_res else None, _stream()))
            dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
            dg2, db2 = torch.empty(C, device=d), torch.empty(C, device=d)
            check(lib.mi355_groupnorm_bwd_cat(x1.data_ptr(), C1, x2.data_ptr(), C2, dy.data_ptr(), stats.data_ptr(),
                                              gamma.data_ptr(), beta.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), dg2.data_ptr(),
                                              db2.data_ptr(), sc.data_ptr(), B, HW, G, silu,
                                              r1.data_ptr() if with_res else None, r2.data_ptr() if with_res else None, _stream()))
            assert torch.equal(dx1, dx[..., :C1]) and torch.equal(dx2, dx[..., C1:])
            assert torch.equal(dg2, dg) and torch.equal(db2, db)
    # one bypass gradient without the other is refused
    assert lib.mi355_groupnorm_bwd_cat(x1.data_ptr(), C1, x2.data_ptr(), C2, dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), None, None, sc.data_ptr(), B, HW, G, 1,
                                       r1.data_ptr(), None, _stream()) != 0


@pytest.mark.parametrize("case", [(512, 320, 320, 320), (2048, 1280, 640, 640), (32768, 640, 320, 320), (300, 160, 8, 48)])
def test_weight_gradient_job_on_two_sources(case):
    """The deferred dense weight-gradient job that reads [x1 | x2] in place (mi355_wgrad_job_fill_cat) against the job on
    the materialised concatenation: equal bit for bit when the job is not pixel-split (same tiles, same order), to
    accumulation order when it is; a C1 that is not a multiple of the 160-channel tile is refused."""
    import ctypes
    from image_generator_amd.ops import lib, WgradQueue
    d = dev()
    M, C1, C2, Co = case
    g = torch.Generator().manual_seed(M + C1)
    x1 = torch.randn(M, C1, generator=g).to(BF).to(d)
    x2 = torch.randn(M, C2, generator=g).to(BF).to(d)
    dy = torch.randn(M, Co, generator=g).to(BF).to(d)
    xc = torch.cat([x1, x2], -1)
    want, got = torch.zeros(Co, C1 + C2, device=d), torch.zeros(Co, C1 + C2, device=d)
    dbw, dbg = torch.zeros(Co, device=d), torch.zeros(Co, device=d)
    q = WgradQueue(d)
    q.push(dy, xc, want, dbw, M, 1, 1, C1 + C2, Co, 1, 1, False, 1, 1)
    split = q._splits.value > 1
    q.push_cat(dy, x1, x2, got, dbg)
    assert (q._splits.value > 1) == split
    q.join()
    torch.cuda.synchronize()
    assert float(want.abs().max()) > 0
    if split:
        close(got, want, 1e-4, 1e-4 * float(want.abs().max()), "split two-source wgrad")
    else:
        assert torch.equal(got, want)
    close(dbg, dbw, 1e-4, 1e-3, "two-source wgrad bias gradient")
    ref = dy.float().t() @ xc.float()
    close(got, ref, 2e-2, 2e-2 * float(ref.abs().max()), "two-source wgrad vs fp32")
    host = (ctypes.c_char * int(lib.mi355_wgrad_job_bytes()))()
    assert lib.mi355_wgrad_job_fill_cat(host, dy.data_ptr(), x1.data_ptr(), 80, x2.data_ptr(), C2, got.data_ptr(), None, M, Co, 0,
                                        None, None, None) < 0


def test_weight_gradient_launch_sorted_by_workgroup_length():
    """mi355_wgrad_jobs_sort puts the jobs with the longest workgroups first and renumbers their blocks: a grouped launch
    of mixed layers (3x3 and dense, unsplit and pixel-split, short and long pixel ranges) gives the gradients of the
    launch in queue order bit for bit where a job is unsplit, to accumulation order where it is split, and the records
    come out ordered."""
    import ctypes
    import numpy as np
    from image_generator_amd.ops import lib, WgradQueue
    d = dev()
    g = torch.Generator().manual_seed(11)
    jobs = [(2, 4, 64, 64, 3), (32, 32, 40, 48, 3), (4, 8, 160, 320, 1), (32, 16, 64, 32, 3), (1, 16, 320, 160, 1), (40, 32, 16, 24, 1)]
    ops_in = []
    for (B, H, ci, co, k) in jobs:
        ops_in.append((torch.randn(B, H, H, ci, generator=g).to(BF).to(d), torch.randn(B, H, H, co, generator=g).to(BF).to(d)))
    outs = {}
    for sort in (False, True):
        q = WgradQueue(d, flush_blocks=1 << 30)
        q.sort_jobs = sort
        res, splits = [], []
        for (B, H, ci, co, k), (x, dy) in zip(jobs, ops_in):
            dw, db = torch.zeros(co, k * k, ci, device=d), torch.zeros(co, device=d)
            q.push(dy.view(-1, co), x, dw, db, B, H, H, ci, co, k, 1, False, H, H)
            splits.append(q._splits.value)
            res.append((dw, db))
        if sort:        # the records as the launch will see them
            jb = int(lib.mi355_wgrad_job_bytes())
            total = lib.mi355_wgrad_jobs_sort(q._host.data_ptr(), q._n)
            assert total == q._blocks
            raw = q._host[:q._n * jb].numpy().reshape(q._n, jb)
            first = raw[:, jb - 8:jb - 4].copy().view(np.int32).ravel()
            nblk = raw[:, jb - 4:].copy().view(np.int32).ravel()
            assert first[0] == 0 and (first[1:] == np.cumsum(nblk)[:-1]).all()
        q.join()
        torch.cuda.synchronize()
        outs[sort] = (res, splits)
    assert outs[False][1] == outs[True][1] and max(outs[True][1]) > 1 and min(outs[True][1]) == 1
    for (a, da), (b, db), sp, (B, H, ci, co, k), (x, dy) in zip(outs[False][0], outs[True][0], outs[True][1], jobs, ops_in):
        assert float(a.abs().max()) > 0
        if sp == 1:
            assert torch.equal(a, b)
        else:
            close(b, a, 1e-4, 1e-4 * float(a.abs().max()), "sorted split wgrad")
        close(db, da, 1e-4, 1e-3, "sorted wgrad bias gradient")
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (co, ci, k, k), dy.float().permute(0, 3, 1, 2),
                                          padding=k // 2).permute(0, 2, 3, 1).reshape(co, k * k, ci)
        close(b, ref, 2e-2, 2e-2 * float(ref.abs().max()), "sorted wgrad vs fp32")


def test_up_path_resnet_trains_without_the_concatenation():
    """ResnetBlock.forward_cat_train (ops.CatNormShortcutFn: two-source normalisation forward and backward, split shortcut,
    one two-source weight-gradient job) against the block on torch.cat([hidden, skip]) in the trainer's gradient mode
    (gradients written into preassigned buffers through the deferred queues): output, both input gradients and every
    parameter gradient."""
    import image_generator_amd.unet as U
    from image_generator_amd import ops
    d = dev()
    torch.manual_seed(5)
    blk = U.ResnetBlock(640 + 320, 320, 1280, 32, 1e-5).to(d)
    layers = [m for m in blk.modules() if isinstance(m, (U.HipConv2d, U.HipLinear))]
    norms = [m for m in blk.modules() if isinstance(m, U.HipGroupNorm)]
    for m in layers:        # a stand-alone block has no flat storage: operand copies and gradient views as the model makes them
        pk = m.pack
        if isinstance(m, U.HipConv2d):
            t = pk.ksize * pk.ksize
            pk.w16 = m.weight.data.permute(0, 2, 3, 1).reshape(pk.cout, t, pk.cin).to(BF).contiguous()
            pk.w16_t = pk.w16.view(pk.cout, t * pk.cin).t().contiguous() if t == 1 else \
                m.weight.data.flip(2, 3).permute(1, 2, 3, 0).reshape(pk.cin, t, pk.cout).to(BF).contiguous()
        else:
            pk.w16 = m.weight.data.to(BF).contiguous()
            pk.w16_t = pk.w16.t().contiguous()
    x1 = rnd(2, 16, 16, 640, seed=51).to(d)
    x2 = rnd(2, 16, 16, 320, seed=52).to(d)
    temb = rnd(2, 1280, seed=53).to(d)
    dout = rnd(2, 16, 16, 320, seed=54).to(d)

    def run(fused):
        for m in layers:
            m.pack.grad_out = torch.zeros(m.pack.n, m.pack.k, device=d)
            m.pack.bias_grad_out = torch.zeros(m.pack.n, device=d)
        for n in norms:
            n.grad_dst = (torch.zeros_like(n.weight), torch.zeros_like(n.bias))
        q, rq = ops.WgradQueue(d), ops.ReduceQueue(d)
        ops.set_wgrad_queue(q)
        ops.set_reduce_queue(rq)
        try:
            a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            if fused:
                assert blk.cat_train_ready(a, b)
                out = blk.forward_cat_train(a, b, temb)
            else:
                out = blk(torch.cat([a, b], -1), temb)
            out.backward(dout)
            q.join()
            rq.flush()
            torch.cuda.synchronize()
        finally:
            ops.set_wgrad_queue(None)
            ops.set_reduce_queue(None)
        grads = [m.pack.grad_out for m in layers] + [m.pack.bias_grad_out for m in layers] + [t for n in norms for t in n.grad_dst]
        return out.detach(), a.grad, b.grad, grads

    o0, a0, b0, g0 = run(False)
    o1, a1, b1, g1 = run(True)
    assert rel_l2(o1, o0.float()) < 4e-3
    assert rel_l2(a1, a0.float()) < 1e-2 and rel_l2(b1, b0.float()) < 1e-2
    for i, (u, v) in enumerate(zip(g1, g0)):
        if float(v.abs().max()) > 0:
            assert rel_l2(u, v) < 1e-2, i
        else:
            assert float(u.abs().max()) == 0, i


@pytest.mark.parametrize("case", [(32768, 640, 320, 320), (8192, 1280, 640, 640), (2048, 1280, 1280, 1280), (512, 1280, 1280, 1280),
                                  (300, 64, 128, 160)])
def test_gemm_on_two_sources_and_into_two_destinations(case):
    """mi355_gemm_cat_bf16 ([a1 | a2] W^T + b with the concatenation read in place) and mi355_gemm_split_bf16 (column
    blocks of a product written as two tensors) against mi355_gemm_bf16 on the materialised concatenation / output:
    the same tiles and k order, so equal bit for bit - with and without split-K (the small-M cases run it)."""
    from image_generator_amd import ops
    d = dev()
    M, K1, K2, N = case
    g = torch.Generator().manual_seed(M + K1 + N)
    a1 = torch.randn(M, K1, generator=g).to(BF).to(d)
    a2 = torch.randn(M, K2, generator=g).to(BF).to(d)
    w = (torch.randn(N, K1 + K2, generator=g) * (K1 + K2) ** -0.5).to(BF).to(d)
    b = torch.randn(N, generator=g).to(d)
    ac = torch.cat([a1, a2], 1)
    want = ops.gemm(ac, w, bias=b)
    got = ops.gemm_cat(a1, a2, w, bias=b)
    assert torch.equal(got, want)
    close(got, ac.float() @ w.float().t() + b, 2e-2, 2e-2 * float(want.float().abs().max()), "gemm_cat vs fp32")
    if K1 % 320 == 0:
        # data-gradient shape of the same layer: [M, N] x W -> [M, K1 | K2]
        dy = torch.randn(M, N, generator=g).to(BF).to(d)
        wt = w.t().contiguous()
        full = ops.gemm(dy, wt)
        o1, o2 = ops.gemm_split(dy, wt, K1)
        assert torch.equal(o1, full[:, :K1]) and torch.equal(o2, full[:, K1:])
    from image_generator_amd.ops import lib
    ws = ops.workspace(d)
    assert lib.mi355_gemm_cat_bf16(a1.data_ptr(), 40, a2.data_ptr(), K2, w.data_ptr(), K1 + K2, M, N, got.data_ptr(), N, None,
                                   ws.data_ptr(), ws.numel(), None) != 0
    assert lib.mi355_gemm_split_bf16(a1.data_ptr(), K1, w.data_ptr(), K1 + K2, M, 160, 160, K1, got.data_ptr(), got.data_ptr(),
                                     ws.data_ptr(), ws.numel(), None) != 0


@pytest.mark.parametrize("n2", [160, 480, 64, 192])
def test_gemm_split_never_lets_a_tile_straddle_the_two_destinations(n2):
    """N1 = 320 with N1 + N2 a multiple of 160 runs on 160-column tiles (none straddles N1) and equals the plain product
    bit for bit; any other total width would take 128-column tiles, one of which would cross N1, and is refused."""
    from image_generator_amd import ops
    from image_generator_amd.ops import lib
    d = dev()
    M, K, N1 = 300, 128, 320
    g = torch.Generator().manual_seed(n2)
    a = torch.randn(M, K, generator=g).to(BF).to(d)
    w = (torch.randn(N1 + n2, K, generator=g) * K ** -0.5).to(BF).to(d)
    if (N1 + n2) % 160 == 0:
        full = ops.gemm(a, w)
        o1, o2 = ops.gemm_split(a, w, N1)
        assert torch.equal(o1, full[:, :N1]) and torch.equal(o2, full[:, N1:])
    else:
        o1 = torch.zeros(M, N1, dtype=BF, device=d)
        o2 = torch.zeros(M + 1, n2, dtype=BF, device=d)
        ws = ops.workspace(d)
        rc = lib.mi355_gemm_split_bf16(a.data_ptr(), K, w.data_ptr(), K, M, N1, n2, K, o1.data_ptr(), o2.data_ptr(),
                                       ws.data_ptr(), ws.numel(), None)
        assert rc != 0 and b"160" in lib.mi355_last_error()
        torch.cuda.synchronize()
        assert float(o1.float().abs().max()) == 0 and float(o2.float().abs().max()) == 0


@pytest.mark.parametrize("case", [(32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (512, 1280, 5120), (200, 64, 80)])
def test_geglu_backward_in_the_gemm_epilogue(case):
    """mi355_gemm_geglu_bwd_bf16 (data gradient of the feed-forward output projection with the gate's backward in its
    epilogue) against the two separate steps: the fused form applies the gate's derivative to the fp32 accumulator, the
    separate one to the product rounded to bf16 first - equal to that rounding; and against fp32 math."""
    from image_generator_amd import ops
    from image_generator_amd.ops import lib, check, _stream
    d = dev()
    M, C, F = case
    g = torch.Generator().manual_seed(M + F)
    dout = torch.randn(M, C, generator=g).to(BF).to(d)
    w_t = (torch.randn(F, C, generator=g) * C ** -0.5).to(BF).to(d)          # [F][C]: the projection's weight transposed
    pre = torch.randn(M, 2 * F, generator=g).to(BF).to(d)
    got = ops.gemm_geglu_bwd(dout, w_t, pre)
    dy = ops.gemm(dout, w_t)
    want = torch.empty_like(pre)
    check(lib.mi355_geglu_bwd(pre.data_ptr(), dy.data_ptr(), want.data_ptr(), M, F, _stream()))
    assert rel_l2(got, want.float()) < 4e-3
    dyf = dout.float() @ w_t.float().t()
    h, gt = pre[:, :F].float(), pre[:, F:].float()
    gelu = torch.nn.functional.gelu(gt)
    dgelu = 0.5 * (1 + torch.erf(gt / 2 ** 0.5)) + gt * torch.exp(-0.5 * gt * gt) / (2 * torch.pi) ** 0.5
    ref = torch.cat([dyf * gelu, dyf * h * dgelu], 1)
    assert rel_l2(got, ref) < 6e-3


def test_feed_forward_block_as_one_node():
    """unet.FeedForward in training mode (ops.FeedForwardFn: the gate's backward inside the output projection's
    data-gradient GEMM) against the two separate nodes (MI355_NO_FF_FUSION=1): output bit-equal, input / residual /
    parameter gradients equal to the bf16 rounding of the gated activations' gradient."""
    import os
    import image_generator_amd.unet as U
    d = dev()
    torch.manual_seed(7)
    ff = U.FeedForward(320).to(d)
    for m in ff.modules():
        if isinstance(m, U.HipLinear):
            m.pack.w16 = m.weight.data.to(BF).contiguous()
            m.pack.w16_t = m.pack.w16.t().contiguous()
    x = rnd(2, 256, 320, seed=61).to(d)
    res = rnd(2, 256, 320, seed=62).to(d)
    dout = rnd(2, 256, 320, seed=63).to(d)

    def run(fused):
        if fused:
            os.environ.pop("MI355_NO_FF_FUSION", None)
        else:
            os.environ["MI355_NO_FF_FUSION"] = "1"
        try:
            for p in ff.parameters():
                p.grad = None
            a, r = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
            out = ff(a, r)
            out.backward(dout)
            torch.cuda.synchronize()
            return out.detach(), a.grad, r.grad, [p.grad.clone() for p in ff.parameters()]
        finally:
            os.environ.pop("MI355_NO_FF_FUSION", None)

    o0, a0, r0, g0 = run(False)
    o1, a1, r1, g1 = run(True)
    assert torch.equal(o1, o0) and torch.equal(r1, r0)
    assert rel_l2(a1, a0.float()) < 5e-3
    for u, v in zip(g1, g0):
        assert rel_l2(u, v.float()) < 5e-3


@pytest.mark.parametrize("shape", [(2, 32, 32, 320), (3, 16, 16, 640), (2, 8, 8, 1280), (2, 4, 4, 1280), (2, 8, 8, 2560)])
def test_groupnorm_backward_with_per_sample_column_sums(shape):
    """mi355_groupnorm_bwd_colsum against mi355_groupnorm_bwd plus the separate column-sum pass over its result
    (mi355_colsum_bf16 with one segment per sample): dx, dgamma, dbeta bit-equal, the sums equal up to fp32 summation
    order - strip kernels and small-map kernel, with and without a bypass gradient."""
    from image_generator_amd.ops import lib, check, _stream, colsum
    d = dev()
    B, H, W, C = shape
    G, HW = 32, H * W
    g = torch.Generator().manual_seed(sum(shape) + 3)
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).to(BF).to(d)
    dy = torch.randn(B, H, W, C, generator=g).to(BF).to(d)
    res = torch.randn(B, H, W, C, generator=g).to(BF).to(d)
    gamma = (torch.randn(C, generator=g) * 0.2 + 1).to(d)
    beta = (torch.randn(C, generator=g) * 0.1).to(d)
    nsc = int(lib.mi355_groupnorm_scratch_floats(B, HW, C, G))
    y = torch.empty_like(x)
    stats = torch.empty(B, G, 2, device=d)
    sc, sc2 = torch.empty(nsc, device=d), torch.empty(nsc, device=d)
    check(lib.mi355_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), sc.data_ptr(),
                                  B, HW, C, G, 1e-5, 1, _stream()))
    for with_res in (False, True):
        rp = res.data_ptr() if with_res else None
        dx0, dg0, db0 = torch.empty_like(x), torch.empty(C, device=d), torch.empty(C, device=d)
        check(lib.mi355_groupnorm_bwd(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      dx0.data_ptr(), dg0.data_ptr(), db0.data_ptr(), sc.data_ptr(), B, HW, C, G, 1, rp, _stream()))
        want = colsum(dx0.view(-1, C), seg=HW)
        dx1, dg1, db1 = torch.empty_like(x), torch.empty(C, device=d), torch.empty(C, device=d)
        sums = torch.full((B, C), 7.0, device=d)
        check(lib.mi355_groupnorm_bwd_colsum(x.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             dx1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), sc.data_ptr(), B, HW, C, G, 1, rp,
                                             sums.data_ptr(), sc2.data_ptr(), _stream()))
        assert torch.equal(dx1, dx0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)
        close(sums, want, 1e-5, 1e-4 * float(want.abs().max()) + 1e-5, "per-sample column sums of dx")
        close(sums, dx0.float().view(B, HW, C).sum(1), 1e-5, 1e-4 * float(want.abs().max()) + 1e-5, "sums vs torch")


def test_resnet_time_embedding_gradient_from_the_norm_backward():
    """ResnetBlock backward with the first convolution's per-sample bias gradient taken from the second normalisation's
    backward kernel (ops._DX_COLSUMS) against the separate column-sum pass: gradients of the timestep embedding, of its
    projection and of the block input - a 16x16 map (strip kernels) and an 8x8 map of 1280 channels (small-map kernel)."""
    import image_generator_amd.unet as U
    from image_generator_amd import ops
    d = dev()
    for (cin, cout, hw) in ((640, 640, 16), (1280, 1280, 8)):
        torch.manual_seed(cin + hw)
        blk = U.ResnetBlock(cin, cout, 1280, 32, 1e-5).to(d)
        for m in blk.modules():
            if isinstance(m, (U.HipConv2d, U.HipLinear)):
                pk = m.pack
                if isinstance(m, U.HipConv2d):
                    t = pk.ksize * pk.ksize
                    pk.w16 = m.weight.data.permute(0, 2, 3, 1).reshape(pk.cout, t, pk.cin).to(BF).contiguous()
                    pk.w16_t = m.weight.data.flip(2, 3).permute(1, 2, 3, 0).reshape(pk.cin, t, pk.cout).to(BF).contiguous()
                else:
                    pk.w16 = m.weight.data.to(BF).contiguous()
                    pk.w16_t = pk.w16.t().contiguous()
        x = rnd(2, hw, hw, cin, seed=71).to(d)
        temb = rnd(2, 1280, seed=72).to(d)
        dout = rnd(2, hw, hw, cout, seed=73).to(d)

        def run(fused):
            old = U._COLSUM_FUSION
            U._COLSUM_FUSION = fused
            try:
                for p in blk.parameters():
                    p.grad = None
                a, t = x.clone().requires_grad_(True), temb.clone().requires_grad_(True)
                blk(a, t).backward(dout)
                torch.cuda.synchronize()
                assert not ops._DX_COLSUMS          # consumed by the convolution's backward
                return a.grad, t.grad, blk.time_emb_proj.weight.grad.clone(), blk.time_emb_proj.bias.grad.clone()
            finally:
                U._COLSUM_FUSION = old

        ref, got = run(False), run(True)
        assert torch.equal(got[0], ref[0])
        for u, v in zip(got[1:], ref[1:]):
            assert rel_l2(u.float(), v.float()) < 2e-3


@pytest.mark.parametrize("M,Nout,Kin", [(300, 128, 160), (2048, 1280, 1280), (8192, 1920, 640), (32, 1280, 1280), (512, 64, 320)])
def test_linear_data_gradient_from_the_weight_as_stored(M, Nout, Kin):
    """mi355_gemm_bt_bf16: dx = dy W with W [out][in] read in place (k-major operand, transposing LDS reads) against fp32
    math, and bit for bit against the same product on a transposed copy of W (same k order, same accumulators)."""
    from image_generator_amd import ops
    d = dev()
    dy = rnd(M, Nout, seed=81).to(d)
    w = rnd(Nout, Kin, scale=Nout ** -0.5, seed=82).to(d)
    got = ops.gemm_bt(dy, w)
    ref = dy.float() @ w.float()
    close(got, ref, 2 ** -7, 2e-2, f"gemm_bt {M}x{Kin}x{Nout}")
    assert rel_l2(got, ref) < 4e-3
    assert torch.equal(got, ops.gemm(dy, w.t().contiguous()))


def test_linear_data_gradient_as_stored_split_and_gated():
    """The two epilogue options of mi355_gemm_bt_bf16 against the entries that take a transposed copy: the column-split
    output (shortcut of an up-path residual block) and the GEGLU backward (feed-forward output projection)."""
    from image_generator_amd import ops
    d = dev()
    M, Nout, C1, C2 = 512, 1280, 1280, 640
    dy = rnd(M, Nout, seed=83).to(d)
    w = rnd(Nout, C1 + C2, scale=Nout ** -0.5, seed=84).to(d)
    d1, d2 = ops.gemm_bt(dy, w, n1=C1)
    e1, e2 = ops.gemm_split(dy, w.t().contiguous(), C1)
    assert d1.shape == (M, C1) and d2.shape == (M, C2) and torch.equal(d1, e1) and torch.equal(d2, e2)
    ref = dy.float() @ w.float()
    assert rel_l2(torch.cat([d1, d2], 1), ref) < 4e-3
    M, C, Fh = 1024, 640, 2560
    dout = rnd(M, C, seed=85).to(d)
    w2 = rnd(C, Fh, scale=C ** -0.5, seed=86).to(d)             # the projection's weight [out = C][in = F]
    pre = rnd(M, 2 * Fh, seed=87).to(d)
    got = ops.gemm_bt(dout, w2, pre=pre)
    assert torch.equal(got, ops.gemm_geglu_bwd(dout, w2.t().contiguous(), pre))


@pytest.mark.parametrize("case", [(2, 16, 16, 128, 160, 3, False, None), (2, 8, 8, 1280, 640, 3, False, None),
                                  (3, 9, 7, 64, 320, 1, False, None), (2, 8, 8, 128, 160, 3, True, (16, 16)),
                                  (2, 8, 8, 64, 160, 3, True, (15, 15)), (32, 4, 4, 1280, 1280, 3, False, None)])
def test_conv_data_gradient_from_the_weight_as_stored(case):
    """mi355_conv2d_dgrad_bt_bf16 (weight [Cout][k][k][Cin] read in place, taps reversed by the addressing) against
    autograd of an fp32 convolution on the same bf16-rounded operands, and bit for bit against the convolution with the
    transposed, tap-flipped weight copy.  Stride-2 layers (dil) with even and odd input sizes."""
    from image_generator_amd import ops
    d = dev()
    B, H, W, Co, Ci, k, dil, out_hw = case
    g = torch.Generator().manual_seed(B * 1000 + Co + Ci + k)
    dy = torch.randn(B, H, W, Co, generator=g).to(BF).to(d)
    w = (torch.randn(Co, k, k, Ci, generator=g) * (k * k * Co) ** -0.5).to(BF).to(d)
    got = ops.conv2d_dgrad_bt(dy, w.view(Co, k * k, Ci), Ci, k, dil=dil, out_hw=out_hw)
    wt = w.flip(1, 2).permute(3, 1, 2, 0).reshape(Ci, k * k, Co).contiguous()
    assert torch.equal(got, ops.conv2d(dy, wt, Ci, k, stride=1, dil=dil, out_hw=out_hw))
    Hi, Wi = out_hw if out_hw is not None else (H, W)
    x = torch.zeros(B, Ci, Hi, Wi, device=d, requires_grad=True)
    y = F.conv2d(x, w.float().permute(0, 3, 1, 2), stride=2 if dil else 1, padding=k // 2)
    assert y.shape == (B, Co, H, W)
    (ref,) = torch.autograd.grad(y, x, dy.float().permute(0, 3, 1, 2))
    assert rel_l2(got, ref.permute(0, 2, 3, 1)) < 4e-3


def test_model_trains_without_transposed_weight_copies():
    """The denoiser with data gradients from the weights as stored against the same model on per-step transposed copies
    (MI355_NO_BT_DGRAD=1 at import: here the module switch): the data gradients are the same products in the same order,
    so the parameter gradients agree to the fp32 summation order of the pixel-split weight-gradient jobs (atomics);
    and the eligible packs really hold no transposed copy."""
    import image_generator_amd.unet as U
    d = dev()
    cfg = U.DenoiserConfig(block_out_channels=(640, 1280), layers_per_block=1, down_has_attn=(True, False), num_heads=8,
                           cross_attention_dim=768, norm_num_groups=32, sample_size=16)
    x = rnd(2, 4, 16, 16, seed=91).to(d).float()
    t = torch.tensor([10, 500], device=d)
    ctx = rnd(2, 77, 768, seed=92).to(d)
    grads = {}
    for bt in (True, False):
        U._BT_DGRAD = bt
        try:
            torch.manual_seed(11)
            net = U.UNet2DConditionModel(cfg).to(d)
            net.train()
            out = net(x, t, encoder_hidden_states=ctx).sample
            out.float().square().mean().backward()
            nbt = sum(1 for _, _, pk, _ in net._packs if pk.bt)
            assert (nbt > 10 and all(pk.w16_t is None for _, _, pk, _ in net._packs if pk.bt)) if bt else nbt == 0
            grads[bt] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            U._BT_DGRAD = True
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 50
    for n in grads[True]:
        assert rel_l2(grads[True][n], grads[False][n]) < 1e-5, n


@pytest.mark.parametrize("case", [(32768, 640, 640, "pp"), (32768, 320, 960, "wide"), (16384, 320, 960, "dma4"),
                                  (8192, 1280, 1280, "dma8"), (2048, 1280, 1280, "dma8x3"), (300, 320, 320, "ragged")])
def test_layernorm_folded_into_the_products_around_it(case):
    """mi355_gemm_rowstats_bf16 -> mi355_gemm_ln_bf16 (plain and gated) against the three separate steps (GEMM with
    residual, LayerNorm kernel, GEMM / gated GEMM) and against fp32 math, on shapes that take each tile kernel.
    (1) the emitting product stores the same bits as mi355_gemm_bf16 and its partial sums add up to the row sums of
    those bits; (2) the folded product equals LayerNorm -> product to bf16 rounding of the normalised activations
    (the fold never rounds them): relative L2 <= 1e-2 against the kernels, <= 1.5e-2 against fp32."""
    from image_generator_amd import ops
    d = dev()
    M, C, N, _ = case
    g = torch.Generator().manual_seed(M + C + N)
    a = torch.randn(M, C, generator=g).to(BF).to(d)
    w0 = (torch.randn(C, C, generator=g) * C ** -0.5).to(BF).to(d)
    b0 = torch.randn(C, generator=g).to(d)
    res = (torch.randn(M, C, generator=g) * 2 + 0.7).to(BF).to(d)          # rows with a mean: the fold subtracts it
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(d), (0.3 * torch.randn(C, generator=g)).to(d)
    w1 = (torch.randn(N, C, generator=g) * C ** -0.5).to(d)
    b1 = torch.randn(N, generator=g).to(d)
    # (1) producer
    x_ref = ops.gemm(a, w0, bias=b0, residual=res)
    x, stat, parts = ops.gemm_rowstats(a, w0, bias=b0, residual=res)
    assert torch.equal(x, x_ref) and parts in (C // 160, C // 320)
    assert stat.shape == (M, parts, 2)
    sums = stat.sum(1)
    xf = x.float()
    assert torch.allclose(sums[:, 0], xf.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[:, 1], (xf * xf).sum(1), rtol=1e-4, atol=1e-2)
    # (2) consumer, plain
    wp, cs, bo = ops.ln_fold_weight(w1, gamma, beta, b1)
    assert torch.equal(wp, (w1 * gamma).to(BF)) and torch.allclose(cs, wp.float().sum(1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(bo, w1 @ beta + b1, rtol=1e-4, atol=1e-4)
    y = ops.gemm_ln(x, wp, cs, bo, stat, parts, 1e-5)
    y_k = ops.gemm(ops.layernorm_bf16(x, gamma, beta, 1e-5), w1.to(BF), bias=b1)
    y_f = torch.nn.functional.layer_norm(xf, (C,), gamma, beta, 1e-5) @ w1.t() + b1
    assert rel_l2(y, y_k) <= 1e-2 and rel_l2(y, y_f) <= 1.5e-2, (rel_l2(y, y_k), rel_l2(y, y_f))
    assert rel_l2(y, y_f) <= 1.5 * rel_l2(y_k, y_f) + 1e-3          # the fold is not the less accurate of the two
    # (3) consumer, gated: N = 2F
    if N % 320 == 0:
        F = N // 2
        z = ops.gemm_ln(x, wp, cs, bo, stat, parts, 1e-5, geglu=True)
        z_f = y_f[:, :F] * torch.nn.functional.gelu(y_f[:, F:])
        z_k = ops.gemm_geglu(ops.layernorm_bf16(x, gamma, beta, 1e-5), w1.to(BF), b1)
        assert z.shape == (M, F) and rel_l2(z, z_k) <= 1.5e-2 and rel_l2(z, z_f) <= 2e-2, (rel_l2(z, z_k), rel_l2(z, z_f))


@pytest.mark.parametrize("case", [("conv", 16, 32, 32, 320, 320, "dma"), ("conv", 8, 64, 64, 320, 320, "pp/wide"), ("conv_res", 32, 32, 32, 640, 320, "rowbias+res"),
                                  ("conv_ups", 32, 16, 16, 320, 320, "ups"), ("dense", 4, 32, 32, 320, 320, "1x1+res"), ("dense", 8, 64, 64, 640, 640, "1x1 wide"),
                                  ("split", 2, 16, 16, 1280, 1280, "split-K: none"), ("conv8", 32, 32, 32, 128, 160, "unit 8")])
def test_groupnorm_statistics_out_of_the_producing_epilogue(case):
    """mi355_conv2d_gnstats_bf16 / mi355_gemm_gnstats_bf16 -> mi355_groupnorm_fwd_pre against the plain product followed by
    mi355_groupnorm_fwd.  (1) the emitting launch stores the plain launch's bits; (2) its sums are the sums of those bits
    per (row tile, run of channels); (3) the GroupNorm fed with them equals the two-pass GroupNorm up to the order of the
    fp32 additions (|diff| <= 1 bf16 ulp on almost every element); (4) a launch that cannot emit (split contraction) says so
    and the caller's fallback is the plain path; (5) two sources with different tilings feed one concatenated norm."""
    from image_generator_amd import ops
    d = dev()
    kind, B, H, W, Cin, Cout, _ = case
    unit = 8 if kind == "conv8" else 10
    G = 32 if unit == 10 else 4
    g = torch.Generator().manual_seed(B * H + Cin + Cout)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.5 + 0.2).to(BF).to(d)
    bias = torch.randn(Cout, generator=g).to(d)
    ups = kind == "conv_ups"
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    res = (torch.randn(B, Ho, Wo, Cout, generator=g)).to(BF).to(d) if kind in ("conv_res", "dense") else None
    rb = torch.randn(B, Cout, generator=g).to(d) if kind == "conv_res" else None
    if kind == "dense":
        w = (torch.randn(Cout, Cin, generator=g) * Cin ** -0.5).to(BF).to(d)
        plain = ops.gemm(x.view(-1, Cin), w, bias=bias, residual=res.view(-1, Cout)).view(B, H, W, Cout)
        out = ops.gemm_gnstats(x.view(-1, Cin), w, bias, res.view(-1, Cout), H * W, unit).view(B, H, W, Cout)
    else:
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (9 * Cin) ** -0.5).to(BF).to(d)
        plain = ops.conv2d(x, w, Cout, 3, ups=ups, bias=bias, rowbias=rb, residual=res)
        out = ops.conv2d(x, w, Cout, 3, ups=ups, bias=bias, rowbias=rb, residual=res, gn_unit=unit)
    pre = ops.take_gn_pre()
    assert torch.equal(out, plain)
    if kind == "split":
        assert pre is None                      # (4)
        return
    assert pre is not None, "the launch should have emitted statistics"
    gstat, tiles, u = pre
    rows = Ho * Wo // tiles
    assert u == unit and rows in (128, 256)
    of = out.float().view(B, tiles, rows, Cout // unit, unit)
    got = gstat.view(B, tiles, Cout // unit, 2)
    assert torch.allclose(got[..., 0], of.sum((2, 4)), rtol=1e-4, atol=5e-2)
    assert torch.allclose(got[..., 1], (of * of).sum((2, 4)), rtol=1e-4, atol=5e-2)
    gamma, beta = torch.randn(Cout, generator=g).to(d), torch.randn(Cout, generator=g).to(d)
    y0, y1 = torch.empty_like(out), torch.empty_like(out)
    s0, s1 = torch.empty(B, G, 2, device=d), torch.empty(B, G, 2, device=d)
    ops.groupnorm_forward(out, None, gamma, beta, y0, s0, G, 1e-5, True)
    ops.groupnorm_forward(out, None, gamma, beta, y1, s1, G, 1e-5, True, pre)
    assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-5)
    diff = (y0.float() - y1.float()).abs()
    assert float((diff > 2 ** -7 * y0.float().abs() + 1e-6).float().mean()) < 1e-3 and float(diff.max()) <= 0.1
    if kind == "dense" and Cout == 640:         # (5): [this 640-wide map | a 320-wide one] under one 32-group norm (30 channels per group)
        x2 = (torch.randn(B, H, W, 320, generator=g)).to(BF).to(d)
        w2 = (torch.randn(320, 3, 3, 320, generator=g) * 2880 ** -0.5).to(BF).to(d)
        o2 = ops.conv2d(x2, w2, 320, 3, bias=None, gn_unit=unit)
        pre2 = ops.take_gn_pre()
        assert pre2 is not None
        gc, bc = torch.randn(960, generator=g).to(d), torch.randn(960, generator=g).to(d)
        yc0, yc1 = torch.empty(B, H, W, 960, dtype=BF, device=d), torch.empty(B, H, W, 960, dtype=BF, device=d)
        sc0, sc1 = torch.empty(B, 32, 2, device=d), torch.empty(B, 32, 2, device=d)
        ops.groupnorm_forward(out, o2, gc, bc, yc0, sc0, 32, 1e-5, True)
        ops.groupnorm_forward(out, o2, gc, bc, yc1, sc1, 32, 1e-5, True, pre, pre2)
        assert torch.allclose(sc0, sc1, rtol=2e-5, atol=2e-5)
        assert float((yc0.float() - yc1.float()).abs().max()) <= 0.1
