"""CPU: host-side logic of the product (tokenizer, data-plane indexing, LR schedule, text encoder module,
state-dict compatibility) and the C-ABI library (loads, exports every declared symbol, argument errors)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol():
    from image_generator_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), name
    assert _lib.lib.mi355_abi_version() == 1


def test_library_rejects_bad_arguments_without_touching_the_gpu():
    from image_generator_amd._lib import lib, check, LibraryError
    rc = lib.mi355_gemm_bf16(None, 8, None, 8, 8, 8, 8, None, 8, 0, None, None, 1, 0, None, 0, 1.0, None, 0, None, 0, 0, 0, 0, 0.0,
                             None, 0, None)
    assert rc == -1 and b"null pointer" in lib.mi355_last_error()
    rc = lib.mi355_conv2d_bf16(1, 1, 4, 4, 8, 1, 8, 5, 1, 0, 0, 4, 4, 1, 0, None, None, 0, None, 1.0, None, 0, None)
    assert rc == -3 and b"ksize" in lib.mi355_last_error()
    with pytest.raises(LibraryError):
        check(lib.mi355_attention_fwd(1, 1, 1, 1, None, 1, 8, 16, 16, 200, 8, 8, 8, 8, 1.0, None), "attention")
    assert lib.mi355_groupnorm_scratch_floats(2, 64, 320, 32) > 0
    assert lib.mi355_colsum_scratch_floats(1024, 320, 1024) > 0
    # split-K workspace: 16 fp32 slabs of the output, capped at 256 MiB
    assert lib.mi355_gemm_workspace_bytes(512, 1280) == 16 * 512 * 1280 * 4
    assert lib.mi355_gemm_workspace_bytes(32768, 10240) == 256 << 20


def test_product_fails_loudly_on_cpu_tensors():
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    from image_generator_amd.loss import ContrastiveLoss
    m = UNet2DConditionModel(DenoiserConfig.small())
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 4, 8, 8), torch.tensor([1]), torch.randn(1, 4, 64))
    with pytest.raises(RuntimeError):
        ContrastiveLoss()(torch.randn(4, 8), torch.randn(4, 8))
    # shapes the level structure cannot carry are named, not left to a concatenation error deep in the up path
    with pytest.raises(ValueError, match="multiples of 4"):
        m(torch.randn(1, 4, 10, 8), torch.tensor([1]), torch.randn(1, 4, 64))
    with pytest.raises(ValueError, match="encoder_hidden_states"):
        m(torch.randn(1, 4, 8, 8), torch.tensor([1]), torch.randn(1, 4, 32))


def test_denoiser_state_dict_is_checkpoint_compatible():
    from oracle.denoiser_ref import DenoiserRef, DenoiserConfig as RC
    from image_generator_amd.unet import UNet2DConditionModel, DenoiserConfig
    ref, hip = DenoiserRef(RC.small()), UNet2DConditionModel(DenoiserConfig.small())
    sr, sh = ref.state_dict(), hip.state_dict()
    assert list(sr.keys()) == list(sh.keys())
    assert all(sr[k].shape == sh[k].shape for k in sr)
    hip.load_state_dict(sr)
    for k in sr:
        assert torch.equal(hip.state_dict()[k], sr[k])
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight" in sh
    assert "up_blocks.1.upsamplers.0.conv.weight" in sh and "mid_block.resnets.1.time_emb_proj.bias" in sh
    # conv weights keep the OIHW logical shape with [O][kh][kw][I] memory
    w = hip.down_blocks[0].resnets[0].conv1.weight
    assert w.shape[2:] == (3, 3) and w.stride()[1] == 1


def test_text_encoder_module_is_state_dict_compatible_and_has_no_cpu_path(golden_dir):
    """The module loads the reference module's state dict unchanged (same keys, shapes, parameter count at the default
    configuration) and computes nothing off the HIP device: CPU tensors raise, as does a geometry the kernels do not
    carry.  (Its arithmetic is compared with the reference-minted vectors in the gpu-marked tests; the CPU restatement
    is compared with the same vectors in test_oracle_golden.py.)"""
    from image_generator_amd.text_encoder import TextEncoder
    z = np.load(os.path.join(golden_dir, "text_encoder.npz"))
    m = TextEncoder(vocab_size=200, embed_dim=32, num_layers=2, n_head=4, projection_dim=24, dropout=0.2)
    m.load_state_dict({k[4:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd__")})
    meta = json.load(open(os.path.join(golden_dir, "text_encoder_meta.json")))
    assert len(TextEncoder().state_dict()) == meta["n_keys"]
    assert sum(p.numel() for p in TextEncoder().parameters()) == meta["default_param_count"]
    ids, mask = torch.tensor(z["ids"]), torch.tensor(z["mask"])
    m.eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        m(ids, mask, return_pooled=False)
    # right-padded masks pass the layout check, any other layout marks its sample
    kv, bad = m._key_counts(mask)
    assert kv.tolist() == mask.sum(1).tolist() and not bool(bad.any())
    holed = mask.clone()
    holed[1, 2] = 0
    left = torch.flip(mask, dims=[1])
    empty = torch.zeros_like(mask)
    assert m._key_counts(holed)[1].tolist() == [False, True, False]
    assert m._key_counts(left)[1].tolist() == [bool(r.sum() < len(r)) for r in mask]
    assert m._key_counts(empty)[1].all() and m._key_counts(empty)[0].tolist() == [1, 1, 1]
    y = m._poison(torch.ones(3, 9, 24), m._key_counts(holed)[1])
    assert torch.isnan(y[1]).all() and torch.isfinite(y[0]).all() and torch.isfinite(y[2]).all()


def test_tokenizer_ids_bit_exact(golden_dir):
    from image_generator_amd.tokenizer import load_tokenizer, HashTokenizer
    d = json.load(open(os.path.join(golden_dir, "tokenizer_cases.json")))
    tok = load_tokenizer(os.path.join(golden_dir, "tokenizer_tiny.json"))
    e = tok(d["texts"], padding="max_length", max_length=77, truncation=True, return_tensors="pt")
    assert e.input_ids.tolist() == d["ids77"] and e.attention_mask.tolist() == d["mask77"]
    e = tok(d["texts"][:3], padding=True, max_length=128, truncation=True, return_tensors="pt")
    assert e.input_ids.tolist() == d["ids_longest"] and e.attention_mask.tolist() == d["mask_longest"]
    assert (tok.pad_token_id, tok.cls_token_id, tok.sep_token_id) == (d["pad"], d["cls"], d["sep"]) == (0, 1, 2)
    h = HashTokenizer(1000)
    e = h(["1girl blue_archive shiroko_(blue_archive)", "", "x " * 200], padding="max_length", max_length=77)
    assert e.input_ids.shape == (3, 77) and e.input_ids[0, 0] == 1 and e.input_ids[1, 1] == 2
    assert int(e.attention_mask[0].sum()) == 8 and int(e.attention_mask[2].sum()) == 77 and e.input_ids[2, 76] == 2
    assert e.input_ids[0, 2] == e.input_ids[0, 5]            # the same token hashes to the same id
    assert h(["a b c", "a"], padding="longest").input_ids.shape == (2, 5)


def test_dataplane_indexing_bit_exact(golden_dir):
    from image_generator_amd import dataplane as dp
    h = json.load(open(os.path.join(golden_dir, "host_logic.json")))
    for c in h["sampler"]:
        assert dp.distributed_sampler_indices(c["n"], c["world"], c["rank"], c["epoch"], 0, c["drop_last"]) == c["idx"]
    assert dp.lmdb_keys(123456) == (b"image-00123456", b"meta-00123456")
    assert dp.lmdb_length(20001) == 10000 and dp.next_index_on_decode_failure(9, 10) == 0
    z = np.load(os.path.join(golden_dir, "collate.npz"))
    r = dp.skip_broken_collate_fn([{"image": torch.tensor(z["a"]), "text": "a b"}, None, {"image": torch.tensor(z["b"]), "text": "c"}])
    assert torch.equal(r["image"], torch.tensor(z["r1_images"])) and r["text"] == ["a b", "c"]
    assert dp.skip_broken_collate_fn([None]) is None


def test_lr_schedule_matches_torch(golden_dir):
    from image_generator_amd.trainer import cosine_warm_restarts_lr
    h = json.load(open(os.path.join(golden_dir, "host_logic.json")))
    for e, lr in h["lr"]:
        assert abs(cosine_warm_restarts_lr(1e-4, e) - lr) <= 1e-12 + 1e-9 * lr


def test_ddim_schedule_host_side():
    from image_generator_amd.schedulers import DDIMScheduler, DDPMScheduler, sd_alphas_cumprod
    from oracle import ops_ref
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == ops_ref.ddim_timesteps(50).tolist()
    assert torch.equal(sd_alphas_cumprod(), ops_ref.sd_alphas_cumprod())
    a_t, a_p, g = s.coefficients(1, 7.5)
    assert a_p == float(s.alphas_cumprod[0]) and g == 7.5
    assert DDPMScheduler().config.num_train_timesteps == 1000
    x = torch.randn(2, 4, 4, 4)
    assert s.scale_model_input(x, 3) is x


def _make_jpeg(seed, w=48, h=40):
    import io
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 255, (h // 8, w // 8, 3), dtype=np.uint8)
    img = Image.fromarray(base, "RGB").resize((w, h), Image.BILINEAR)        # smooth content: JPEG keeps it recognisable
    buf = io.BytesIO()
    img.save(buf, format="JPEG", quality=95)
    return buf.getvalue(), img


def test_lmdb_reader_and_dataset(tmp_path):
    """The data plane of dataset.py:169-229 on the library's LMDB page walker: a data base in liblmdb's file layout
    (written by tests/lmdb_fixture.py: 1 200 samples = 2 400 keys, JPEG values on overflow pages, pickled metadata inline,
    a three-level tree, the stale meta page first) is read back key by key, bit for bit; the dataset returns the
    reference's sample dictionary; a corrupt image falls through to the next index; metadata that would import code is
    refused.  (Format parity is unpinned: no file written by liblmdb itself exists here.)"""
    import pickle
    from lmdb_fixture import write_lmdb
    from image_generator_amd import dataplane as dp
    n = 1200
    items, jpegs = {}, {}
    for i in range(n):
        ki, km = dp.lmdb_keys(i)
        if i < 6:
            jpegs[i] = _make_jpeg(i)
            items[ki] = jpegs[i][0]
        else:
            items[ki] = bytes([i & 255]) * (3000 + 7 * i)                    # overflow values of many sizes
        items[km] = pickle.dumps({"raw_text": f"tag_{i} blue_sky"})
    items[dp.lmdb_keys(3)[0]] = b"not a jpeg" * 400                          # sample 3 does not decode
    d = tmp_path / "db"
    d.mkdir()
    info = write_lmdb(str(d / "data.mdb"), items, txnid=7, fill=0.12)          # sparse pages: a deeper tree from few keys
    assert info["depth"] >= 3 and info["overflow"] > n
    env = dp.LMDBEnv(d)
    assert env.entries == 2 * n
    assert int(env._lib.mi355_lmdb_depth(env._env)) == info["depth"] and int(env._lib.mi355_lmdb_page_size(env._env)) == 4096
    for k, v in items.items():
        assert env.get(k) == v, k
    assert env.get(b"image-99999999") is None and env.get(b"") is None and env.get(b"zzz") is None
    env.close()
    # a torn meta page 0 (here: its magic overwritten): the environment opens on meta page 1, as liblmdb would
    torn = tmp_path / "torn"
    torn.mkdir()
    blob = bytearray((d / "data.mdb").read_bytes())
    live = 0 if env_live_meta(blob) == 0 else 1
    if live == 0:                # make page 1 the live one by swapping the two meta pages' bodies (pgno fields aside)
        blob[16:4096], blob[4096 + 16:8192] = blob[4096 + 16:8192], blob[16:4096]
    blob[16:20] = b"\0\0\0\0"
    (torn / "data.mdb").write_bytes(bytes(blob))
    env = dp.LMDBEnv(torn)
    assert env.entries == 2 * n and env.get(dp.lmdb_keys(17)[1]) == items[dp.lmdb_keys(17)[1]]
    env.close()
    with pytest.raises(OSError):
        dp.LMDBEnv(tmp_path / "missing")
    (tmp_path / "junk").write_bytes(b"\0" * 8192)
    with pytest.raises(OSError, match="not an LMDB"):
        dp.LMDBEnv(tmp_path / "junk")

    ds = dp.LMDBImageDataset(d, is_train=False)
    assert len(ds) == n
    s = ds[1]
    assert set(s) == {"image", "text"} and s["text"] == {"raw_text": "tag_1 blue_sky"}
    assert s["image"].shape == (3, 224, 224) and s["image"].dtype == torch.float32
    want = torch.from_numpy(np.asarray(Image_open_rgb(jpegs[1][0]).resize((224, 224), 2), dtype=np.uint8).copy())
    assert torch.equal(s["image"], (want.permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5)
    assert ds[3]["text"]["raw_text"] == "tag_4 blue_sky"                      # corrupt image: the next sample stands in
    tr = dp.LMDBImageDataset(d, is_train=True)
    torch.manual_seed(0)
    a = tr[2]["image"]
    torch.manual_seed(0)
    b = tr[2]["image"]
    assert a.shape == (3, 512, 512) and torch.equal(a, b) and float(a.min()) >= -1 and float(a.max()) <= 1
    flips = set()
    for seed in range(8):
        torch.manual_seed(seed)
        flips.add(bool(torch.equal(tr[2]["image"], a)))
    assert flips == {True, False}                                            # the flip / jitter draws do vary with the seed
    # a DataLoader worker re-opens the map from the pickled dataset; the collate function is the reference's
    import pickle as pk
    ds2 = pk.loads(pk.dumps(ds))
    batch = dp.skip_broken_collate_fn([ds2[0], None, ds2[1]])
    assert batch["image"].shape == (2, 3, 224, 224) and batch["text"][1] == {"raw_text": "tag_1 blue_sky"}

    class Evil:
        def __reduce__(self):
            return (print, ("executed",))
    with pytest.raises(Exception, match="only plain"):
        dp.load_metadata(pk.dumps({"raw_text": Evil()}))


def env_live_meta(blob):
    """Index of the meta page with the larger transaction id (page size 4096)."""
    import struct
    txn = [struct.unpack_from("<Q", blob, pg * 4096 + 16 + 24 + 48 + 56)[0] for pg in (0, 1)]
    return 0 if txn[0] > txn[1] else 1


def Image_open_rgb(blob):
    import io
    from PIL import Image
    return Image.open(io.BytesIO(blob)).convert("RGB")


def test_bench_refuses_more_ranks_than_visible_gpus_without_touching_a_gpu():
    """`python bench.py --gpus N` with fewer than N GPUs visible (none in a CPU container) exits 2 with a message and
    prints no line; the launching process itself never loads a GPU runtime (the count comes from a child)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch as _t
    n = _t.cuda.device_count() + 1 if _t.cuda.device_count() else 2
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--small"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and '"metric"' not in r.stdout
