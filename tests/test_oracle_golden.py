"""CPU: the oracle restatements (oracle/ops_ref.py) against the golden vectors minted from the
reference's own modules (tests/golden/make_golden.py), and the denoiser oracle's structural pins."""
import json
import os

import numpy as np
import torch

from oracle import ops_ref
from oracle.denoiser_ref import DenoiserRef, DenoiserConfig, count_params, timestep_embedding


def _t(z, k):
    return torch.tensor(z[k])


def test_lora_oracle_matches_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "lora.npz"))
    for ci in range(2):
        x = _t(z, f"c{ci}_x").requires_grad_(True)
        a, b = _t(z, f"c{ci}_a").requires_grad_(True), _t(z, f"c{ci}_bb").requires_grad_(True)
        bias = _t(z, f"c{ci}_b")
        y = ops_ref.lora_linear(x, _t(z, f"c{ci}_w"), bias if bias.numel() else None, a, b, float(z[f"c{ci}_alpha"]))
        y.backward(_t(z, f"c{ci}_go"))
        assert torch.allclose(y, _t(z, f"c{ci}_y"), rtol=1e-5, atol=1e-5)
        assert torch.allclose(x.grad, _t(z, f"c{ci}_dx"), rtol=1e-4, atol=1e-5)
        assert torch.allclose(a.grad, _t(z, f"c{ci}_da"), rtol=1e-4, atol=1e-5)
        assert torch.allclose(b.grad, _t(z, f"c{ci}_db"), rtol=1e-4, atol=1e-5)
        fin = z[f"c{ci}_w"].shape[1]
        assert float(z[f"c{ci}_init_a_absmax"]) <= ops_ref.lora_init_bound(fin) + 1e-7
    keys = json.load(open(os.path.join(golden_dir, "lora_keys.json")))
    assert sorted(keys["keys"]) == ["0.lora_A", "0.lora_B", "2.lora_A", "2.lora_B"] and keys["roundtrip_equal"]
    assert all(v == ["weight"] for v in keys["keys"].values())


def test_contrastive_oracle_matches_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "contrastive.npz"))
    for ci in range(4):
        t, i = _t(z, f"c{ci}_t").requires_grad_(True), _t(z, f"c{ci}_i").requires_grad_(True)
        lt = _t(z, f"c{ci}_log_temp").requires_grad_(True)
        loss = ops_ref.contrastive_loss(t, i, lt)
        loss.backward()
        assert torch.allclose(loss, _t(z, f"c{ci}_loss"), rtol=1e-5, atol=1e-6)
        assert torch.allclose(t.grad, _t(z, f"c{ci}_dt"), rtol=1e-3, atol=1e-6)
        assert torch.allclose(i.grad, _t(z, f"c{ci}_di"), rtol=1e-3, atol=1e-6)
        assert torch.allclose(lt.grad, _t(z, f"c{ci}_dlog_temp"), rtol=1e-3, atol=1e-6)


def test_text_encoder_oracle_matches_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "text_encoder.npz"))
    sd = {k[4:]: _t(z, k) for k in z.files if k.startswith("sd__")}
    ids, mask = _t(z, "ids"), _t(z, "mask")
    for pooled, key in ((False, "seq"), (True, "pooled")):
        ev = ops_ref.text_encoder_forward(sd, ids, mask, 4, pooled, zero_padded_rows=True)
        tr = ops_ref.text_encoder_forward(sd, ids, mask, 4, pooled, zero_padded_rows=False)
        assert torch.allclose(ev, _t(z, "eval_" + key), rtol=1e-5, atol=2e-6)
        assert torch.allclose(tr, _t(z, "train_" + key), rtol=1e-5, atol=2e-6)
    # the two modes differ on padded rows only
    pad = ~mask.bool()
    d = (_t(z, "eval_seq") - _t(z, "train_seq")).abs()
    assert float(d[~pad].max()) < 1e-5 and float(d[pad].max()) > 1e-3


def test_text_encoder_oracle_matches_reference_vectors_at_head_width_96(golden_dir):
    """The same restatement against the reference module at its own head width (192 wide, 2 heads of 96; weights
    regenerated from tests/seeded_weights.py): eval and train sequence outputs, pooled output, and the contrastive step's
    loss and parameter gradients through autograd of the restatement."""
    from seeded_weights import text_encoder_state
    z = np.load(os.path.join(golden_dir, "text_encoder_d96.npz"))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in text_encoder_state(500, 300, 192, 2, 64).items()}
    ids, mask = _t(z, "ids"), _t(z, "mask")
    with torch.no_grad():
        assert torch.allclose(ops_ref.text_encoder_forward(sd, ids, mask, 2, False, True), _t(z, "eval_seq"), rtol=1e-5, atol=5e-6)
        assert torch.allclose(ops_ref.text_encoder_forward(sd, ids, mask, 2, True, True), _t(z, "eval_pooled"), rtol=1e-5, atol=5e-6)
        assert torch.allclose(ops_ref.text_encoder_forward(sd, ids, mask, 2, False, False), _t(z, "train_seq"), rtol=1e-5, atol=5e-6)
    pooled = ops_ref.text_encoder_forward(sd, ids, mask, 2, True, False)
    img, lt = _t(z, "img").requires_grad_(True), _t(z, "log_temp").clone().requires_grad_(True)
    loss = ops_ref.contrastive_loss(pooled, img, lt)
    loss.backward()
    assert torch.allclose(loss, _t(z, "loss"), rtol=1e-5)
    assert torch.allclose(img.grad, _t(z, "d_img"), rtol=1e-3, atol=1e-6)
    for k, v in sd.items():
        want = _t(z, "grad__" + k)
        assert float((v.grad - want).norm()) <= 1e-4 * float(want.norm()) + 1e-7, k


def test_collate_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "collate.npz"))
    meta = json.load(open(os.path.join(golden_dir, "collate.json")))
    a, b = {"image": _t(z, "a"), "text": "a b"}, {"image": _t(z, "b"), "text": "c"}
    r = ops_ref.skip_broken_collate([a, None, b])
    assert sorted(r.keys()) == meta["r1_keys"] and r["text"] == meta["r1_texts"]
    assert torch.equal(r["image"], _t(z, "r1_images"))
    assert ops_ref.skip_broken_collate([None, None]) is None
    assert ops_ref.skip_broken_collate([a, {"image": torch.full((3, 4, 4), float("nan")), "text": "d"}]) is None


def test_host_logic_oracle(golden_dir):
    h = json.load(open(os.path.join(golden_dir, "host_logic.json")))
    for c in h["sampler"]:
        assert ops_ref.distributed_sampler_indices(c["n"], c["world"], c["rank"], c["epoch"], 0, c["drop_last"]) == c["idx"]
    for e, lr in h["lr"]:
        assert abs(ops_ref.cosine_warm_restarts_lr(1e-4, e) - lr) <= 1e-12 + 1e-9 * lr
    assert ops_ref.lmdb_keys(7) == (b"image-00000007", b"meta-00000007") and ops_ref.lmdb_length(11) == 5


def test_denoiser_oracle_structure():
    assert count_params(DenoiserRef(DenoiserConfig.sd_v1())) == 859_520_964      # SD-v1 UNet parameter count
    e = timestep_embedding(torch.tensor([0, 10]), 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))
    m = DenoiserRef(DenoiserConfig.small())
    x, t, c = torch.randn(2, 4, 8, 8), torch.tensor([1, 999]), torch.randn(2, 5, 64)
    y = m(x, t, c).sample
    assert y.shape == x.shape
    # no attention mask: permuting the context tokens leaves the output unchanged
    y2 = m(x, t, c[:, torch.randperm(5)]).sample
    assert torch.allclose(y, y2, atol=1e-5)
    # DDIM/ DDPM identities
    acp = ops_ref.sd_alphas_cumprod()
    assert abs(float(acp[0]) - (1 - 0.00085)) < 1e-6 and float(acp[-1]) < 0.01
    assert list(ops_ref.ddim_timesteps(50)[:3]) == [981, 961, 941] and ops_ref.ddim_timesteps(50)[-1] == 1
    x0, eps = torch.randn(1, 4, 4, 4), torch.randn(1, 4, 4, 4)
    xt = ops_ref.ddpm_add_noise(x0, eps, torch.tensor([481]), acp)
    prev = ops_ref.ddim_step(eps, 481, xt, acp, 50)             # with the true noise DDIM lands on the t-20 marginal
    assert torch.allclose(prev, ops_ref.ddpm_add_noise(x0, eps, torch.tensor([461]), acp), atol=1e-5)


def test_vae_oracle_parameter_count_and_shapes():
    """The auto-encoder restatement has the public SD-v1 checkpoint's parameter count and names (its only external
    pin: parity with the third-party package is unpinned), and the bottom/right-padded down-samplers halve sizes."""
    from oracle.vae_ref import AutoencoderKLRef, VaeConfig, count_params
    with torch.device("meta"):
        full = AutoencoderKLRef(VaeConfig.sd_v1())
    assert count_params(full) == 83_653_863
    keys = set(full.state_dict())
    for k in ("encoder.down_blocks.0.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_q.bias",
              "decoder.up_blocks.3.resnets.2.conv2.weight", "decoder.up_blocks.2.upsamplers.0.conv.bias",
              "quant_conv.weight", "post_quant_conv.bias", "decoder.mid_block.attentions.0.to_out.0.weight"):
        assert k in keys, k
    torch.manual_seed(0)
    small = AutoencoderKLRef(VaeConfig.small()).eval()
    x = torch.rand(1, 3, 32, 48) * 2 - 1
    with torch.no_grad():
        d = small.encode(x).latent_dist
        assert d.mean.shape == (1, 4, 4, 6) and float(d.logvar.max()) <= 20.0
        assert small.decode(d.mode()).sample.shape == x.shape


def test_bf16_storage_noise_floor_of_the_denoiser_oracle():
    """What bf16 storage alone (weights and activations rounded, arithmetic fp32) does to the oracle's own gradients:
    the floor the GPU parity tests scale their thresholds from.  Small denoiser, batch 4, 8x8 latents: every
    parameter-gradient cosine stays above 0.9993 (measured 0.99955), the worst relative L2 below 0.04 (measured
    0.030), the median below 0.02 (measured 0.016), the loss moves by less than 1e-3."""
    from oracle.bf16_emulation import emulate_bf16_storage, gradient_distances
    from oracle.denoiser_ref import DenoiserRef, DenoiserConfig, mse_objective
    torch.manual_seed(0)
    ref = DenoiserRef(DenoiserConfig.small())
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(4, 4, 8, 8, generator=g), torch.randint(0, 1000, (4,), generator=g)
    ctx, noise = torch.randn(4, 16, 64, generator=g), torch.randn(4, 4, 8, 8, generator=g)
    l0 = mse_objective(ref(x, t, ctx).sample, noise)
    l0.backward()
    emu = emulate_bf16_storage(ref)
    l1 = mse_objective(emu(x.to(torch.bfloat16).float(), t, ctx.to(torch.bfloat16).float()).sample, noise)
    l1.backward()
    d = gradient_distances(emu, ref)
    assert len(d) == len(list(ref.parameters()))
    rels = sorted(r for _, r in d.values())
    assert min(c for c, _ in d.values()) >= 0.9993 and rels[-1] <= 0.04 and rels[len(rels) // 2] <= 0.02
    assert abs(float(l1) - float(l0)) <= 1e-3 * float(l0)
    assert rels[-1] >= 0.005            # and the floor is not zero: the hooks do round
