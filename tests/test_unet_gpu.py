"""End-to-end parity of the HIP denoiser against the fp32 CPU oracle (oracle/denoiser_ref.py) on
identical weights, seeds and inputs.

Tolerances: the HIP path keeps weights and activations in bf16 (8 significant bits) with fp32 accumulation.  What that
alone costs is measured, in the same test, by re-running the oracle with bf16-rounded storage
(oracle/bf16_emulation.py); the HIP path must stay within twice that floor on the network output and on the worst and
the median parameter gradient (cosine and relative L2), and every gradient cosine must reach 0.999.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(cfg_name, seed=0):
    from oracle.denoiser_ref import DenoiserRef, DenoiserConfig as RefCfg
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    torch.manual_seed(seed)
    if cfg_name == "small":
        rc, hc = RefCfg.small(), DenoiserConfig.small()
    else:   # SD-like channel widths (320/640: exercises the 160-wide tiles, head dims 40/80) at reduced depth
        kw = dict(block_out_channels=(320, 640), layers_per_block=1, down_has_attn=(True, False), num_heads=8,
                  cross_attention_dim=768, norm_num_groups=32)
        rc, hc = RefCfg(**kw), DenoiserConfig(**kw)
    ref = DenoiserRef(rc)
    # biases/norm affine away from their trivial init so their gradients are exercised
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    hip = UNet2DConditionModel(hc)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.to("cuda:0"), rc


def _inputs(rc, B, hw, T, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, rc.in_channels, hw, hw, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ctx = torch.randn(B, T, rc.cross_attention_dim, generator=g)
    noise = torch.randn(B, rc.in_channels, hw, hw, generator=g)
    return x, t, ctx, noise


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def _cos(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


# ragged cases: one sample, one context token, maps of 12x12 / 20x20 (levels 6x6, 3x3 / 10x10, 5x5: no tile is full)
@pytest.mark.parametrize("cfg_name,B,hw,T", [("small", 4, 8, 16), ("small", 2, 16, 7), ("sdlike", 2, 16, 77),
                                             ("small", 1, 12, 1), ("small", 3, 20, 5)])
def test_unet_forward_backward_parity(cfg_name, B, hw, T):
    """Thresholds come from a measured noise floor, not from a guess: the same oracle is run a second time with the HIP
    path's storage precision (bf16 weights and activations, fp32 arithmetic; oracle/bf16_emulation.py) and its distance
    from the fp32 run is what bf16 storage alone costs - worst parameter-gradient cosine 0.9995-0.9998 and relative L2
    0.02-0.03 on these cases.  The HIP path must stay within TWICE that floor on the worst cosine and the worst relative
    L2 (measured: 1.1-1.8 x) and within 1.5 x on their medians (measured: 1.0-1.1 x), all evaluated in the same run."""
    from oracle.bf16_emulation import emulate_bf16_storage, gradient_distances
    from oracle.denoiser_ref import mse_objective
    from image_generator_amd.ops import mse_loss
    ref, hip, rc = _pair(cfg_name)
    x, t, ctx, noise = _inputs(rc, B, hw, T)
    out_ref = ref(x, t, ctx).sample
    loss_ref = mse_objective(out_ref, noise)
    loss_ref.backward()
    emu = emulate_bf16_storage(ref)
    bf = lambda v: v.to(torch.bfloat16).float()
    out_emu = emu(bf(x), t, bf(ctx)).sample
    mse_objective(out_emu, noise).backward()
    floor = gradient_distances(emu, ref)
    d = "cuda:0"
    out = hip(x.to(d), t.to(d), encoder_hidden_states=ctx.to(d)).sample
    assert out.shape == out_ref.shape and out.dtype == torch.float32
    loss = mse_loss(out, noise.to(d))
    loss.backward()
    out_floor = _rel(out_emu, out_ref)
    assert _rel(out, out_ref) <= 2 * out_floor, (_rel(out, out_ref), out_floor)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 2e-3 * abs(float(loss_ref.detach())), (float(loss.detach()), float(loss_ref.detach()))
    rp = dict(ref.named_parameters())
    for n, p in hip.named_parameters():
        assert p.grad is not None, n
        if float(rp[n].grad.norm()) < 1e-12:
            # one context token: softmax over a single key is 1, so nothing upstream of the cross-attention query has a
            # gradient; the HIP path must produce (numerically) nothing there either
            assert float(p.grad.float().norm()) <= 1e-6, (n, float(p.grad.float().norm()))
    got = gradient_distances(hip, ref)
    assert set(got) == set(floor)
    worst_cos_f, worst_rel_f = min(c for c, _ in floor.values()), max(r for _, r in floor.values())
    med = lambda v: sorted(v)[len(v) // 2]
    med_cos_f, med_rel_f = med([c for c, _ in floor.values()]), med([r for _, r in floor.values()])
    worst_cos, worst_rel = min(got.items(), key=lambda kv: kv[1][0]), max(got.items(), key=lambda kv: kv[1][1])
    med_cos, med_rel = med([c for c, _ in got.values()]), med([r for _, r in got.values()])
    print(f"{cfg_name} B{B} {hw}x{hw} T{T}: output rel-L2 {_rel(out, out_ref):.2e} (floor {out_floor:.2e}); gradient cosine worst "
          f"{worst_cos[1][0]:.5f} (floor {worst_cos_f:.5f}) median {med_cos:.6f} (floor {med_cos_f:.6f}); relative L2 worst "
          f"{worst_rel[1][1]:.3e} (floor {worst_rel_f:.3e}) median {med_rel:.3e} (floor {med_rel_f:.3e})")
    assert 1 - worst_cos[1][0] <= 2 * (1 - worst_cos_f), (worst_cos, worst_cos_f)
    assert worst_rel[1][1] <= 2 * worst_rel_f, (worst_rel, worst_rel_f)
    assert 1 - med_cos <= 1.5 * (1 - med_cos_f) and med_rel <= 1.5 * med_rel_f, (med_cos, med_cos_f, med_rel, med_rel_f)
    assert worst_cos[1][0] >= 0.999          # in absolute terms: every parameter gradient, cosine >= 0.999


def test_unet_direct_grads_match_autograd_grads():
    """Writing weight gradients straight into the flat buffer gives the same numbers as autograd's."""
    from image_generator_amd.ops import mse_loss
    ref, hip, rc = _pair("small")
    x, t, ctx, noise = _inputs(rc, 2, 8, 16)
    d = "cuda:0"
    loss = mse_loss(hip(x.to(d), t.to(d), ctx.to(d)).sample, noise.to(d))
    loss.backward()
    g0 = {n: p.grad.detach().clone() for n, p in hip.named_parameters()}
    for p in hip.parameters():
        p.grad = None
    flat = hip.enable_direct_grads()
    hip.zero_small_grads()
    loss = mse_loss(hip(x.to(d), t.to(d), ctx.to(d)).sample, noise.to(d))
    loss.backward()
    for n, p in hip.named_parameters():
        assert p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4, n
        assert _rel(p.grad, g0[n]) < 1e-3, (n, _rel(p.grad, g0[n]))


def test_unet_state_dict_roundtrip_and_refresh():
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    ref, hip, rc = _pair("small")
    x, t, ctx, _ = _inputs(rc, 2, 8, 16)
    d = "cuda:0"
    with torch.no_grad():
        y0 = hip(x.to(d), t.to(d), ctx.to(d)).sample
        sd = {k: v.clone() for k, v in hip.state_dict().items()}
        for p in hip.parameters():          # in-place update must be picked up by the operand refresh
            p.mul_(0.5)
        y1 = hip(x.to(d), t.to(d), ctx.to(d)).sample
        assert _rel(y1, y0) > 1e-2
        hip2 = UNet2DConditionModel(DenoiserConfig.small()).to(d)
        hip2.load_state_dict(sd)
        y2 = hip2(x.to(d), t.to(d), ctx.to(d)).sample
    assert torch.equal(y2, y0)


@pytest.mark.parametrize("cfg_name,B,hw,T", [("small", 3, 8, 9), ("sdlike", 2, 16, 77)])
def test_text_keys_and_values_of_all_layers_from_one_product(cfg_name, B, hw, T, monkeypatch):
    """The to_k / to_v projections of every cross-attention layer multiply the same text states and run as ONE product
    on a packed operand (ops.ContextKVFn); with MI355_NO_CTX_KV_FUSION=1 each layer projects alone.  Same weights, same
    inputs: outputs and every parameter gradient agree to bf16 rounding (the two forms accumulate in different tile
    orders), the packed operand follows the weights through an optimizer-style update, and under no_grad with the
    sampling loop's cache the product runs once."""
    from image_generator_amd import ops
    from image_generator_amd.unet import context_kv_cache
    from image_generator_amd.ops import mse_loss
    _, fused, rc = _pair(cfg_name)
    monkeypatch.setenv("MI355_NO_CTX_KV_FUSION", "1")
    _, plain, _ = _pair(cfg_name)
    plain._ensure_flat()                    # (the switch is read when the flat storage is laid out)
    monkeypatch.delenv("MI355_NO_CTX_KV_FUSION")
    plain.load_state_dict(fused.state_dict())
    x, t, ctx, noise = (v.cuda() for v in _inputs(rc, B, hw, T))
    outs = []
    for m in (fused, plain):
        m._ensure_flat()
        loss = mse_loss(m(x, t, encoder_hidden_states=ctx).sample, noise)
        loss.backward()
        outs.append(float(loss.detach()))
    assert len(fused._ctx_layers) >= 2 and not plain._ctx_layers
    assert abs(outs[0] - outs[1]) <= 2e-3 * abs(outs[1])
    gp = dict(plain.named_parameters())
    for n, p in fused.named_parameters():
        if ".attn2.to_k." in n or ".attn2.to_v." in n or n.startswith("conv_in"):
            assert _cos(p.grad, gp[n].grad) >= 0.999 and _rel(p.grad, gp[n].grad) <= 3e-2, n
    # the packed operand is rebuilt with the other operand copies when a weight changes
    with torch.no_grad():
        fused._ctx_layers[1].to_v.weight.mul_(1.5)
    fused._maybe_refresh()
    r0 = fused._ctx_layers[0].fused_pack.n
    pk = fused._ctx_layers[1].fused_pack
    assert torch.equal(fused._ctx_w16[r0:r0 + pk.n], pk.w16.view(pk.n, pk.k))
    # sampling loop: one product for the whole loop
    calls = []
    orig = ops.ContextKVFn.apply
    monkeypatch.setattr(ops.ContextKVFn, "apply", staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
    ctx16 = ctx.to(torch.bfloat16).contiguous()
    with torch.no_grad(), context_kv_cache():
        a = fused(x, t, encoder_hidden_states=ctx16).sample
        b = fused(x, t, encoder_hidden_states=ctx16).sample
    assert len(calls) == 1 and torch.equal(a, b)


def test_sampling_call_with_layernorms_folded_into_the_products(monkeypatch):
    """Under no_grad at SD widths the transformer blocks run without LayerNorm launches (the norm is applied in the
    epilogue of the product behind it, its statistics come out of the product in front of it).  Same model, same inputs,
    fold on and off: the two outputs differ by bf16 rounding of the normalised activations only, and both sit at the
    same distance from the fp32 oracle."""
    import image_generator_amd.unet as U
    from image_generator_amd import ops
    ref, hip, rc = _pair("sdlike")
    x, t, ctx, _ = _inputs(rc, 4, 32, 77)
    with torch.no_grad():
        want = ref(x, t, ctx).sample
        calls = []
        orig = ops.gemm_ln
        monkeypatch.setattr(ops, "gemm_ln", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        folded = hip(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
        n_folded = len(calls)
        monkeypatch.setattr(U, "_LN_FOLD", False)
        plain = hip(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
    assert n_folded == 9 and len(calls) == 9            # three transformer blocks at 320 channels x three norms; none when off
    assert _rel(folded, plain) <= 1e-2, _rel(folded, plain)
    assert _rel(folded, want) <= 2e-2 and _rel(folded, want) <= 1.5 * _rel(plain, want) + 1e-3, (_rel(folded, want), _rel(plain, want))
