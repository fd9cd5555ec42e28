"""Mint golden vectors from the reference's own importable modules.

Run ONLY in the build container (needs /root/reference):  python tests/golden/make_golden.py
It imports lora.py, loss.py, encoder/text_encoder.py, utils/collate_fn.py and
tokenizer/train_tokenizer.py from /root/reference, feeds them seeded inputs and stores inputs,
weights and outputs as .npz / .json next to this file.  The outputs are data; no reference source
is copied.  torch-owned host logic (DistributedSampler, CosineAnnealingWarmRestarts) is recorded
from torch itself.
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def npz(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                      for k, v in arrs.items()})


def lora_golden():
    from lora import LoRALinear, extract_lora_weights, load_lora_weights
    out = {}
    for ci, (fin, fout, alpha, bias) in enumerate([(320, 320, 1.0, True), (768, 640, 0.5, False)]):
        torch.manual_seed(100 + ci)
        lin = torch.nn.Linear(fin, fout, bias=bias)
        m = LoRALinear(lin, r=4, alpha=alpha)
        init_a = m.lora_A.weight.detach().clone()
        assert float(m.lora_B.weight.abs().max()) == 0.0
        with torch.no_grad():
            m.lora_B.weight.normal_(0, 0.05)
        x = torch.randn(3, 7, fin, requires_grad=True)
        y = m(x)
        go = torch.randn_like(y)
        y.backward(go)
        assert lin.weight.grad is None
        out.update({f"c{ci}_w": lin.weight, f"c{ci}_b": lin.bias if bias else torch.zeros(0),
                    f"c{ci}_a": m.lora_A.weight, f"c{ci}_bb": m.lora_B.weight, f"c{ci}_x": x, f"c{ci}_y": y,
                    f"c{ci}_go": go, f"c{ci}_dx": x.grad, f"c{ci}_da": m.lora_A.weight.grad,
                    f"c{ci}_db": m.lora_B.weight.grad, f"c{ci}_alpha": torch.tensor(alpha),
                    f"c{ci}_init_a_absmax": init_a.abs().max()})
    npz("lora.npz", **out)
    # key naming of extract/load
    torch.manual_seed(7)
    seq = torch.nn.Sequential(LoRALinear(torch.nn.Linear(8, 8)), torch.nn.ReLU(), LoRALinear(torch.nn.Linear(8, 4)))
    w = extract_lora_weights(seq)
    keys = {k: sorted(v.keys()) for k, v in w.items()}
    with torch.no_grad():
        for m in seq:
            if isinstance(m, LoRALinear):
                m.lora_B.weight.fill_(0.25)
    w2 = extract_lora_weights(seq)
    seq2 = torch.nn.Sequential(LoRALinear(torch.nn.Linear(8, 8)), torch.nn.ReLU(), LoRALinear(torch.nn.Linear(8, 4)))
    load_lora_weights(seq2, w2)
    ok = all(torch.equal(a.lora_B.weight, b.lora_B.weight) and torch.equal(a.lora_A.weight, b.lora_A.weight)
             for a, b in zip(seq, seq2) if isinstance(a, LoRALinear))
    json.dump({"keys": keys, "roundtrip_equal": bool(ok)}, open(os.path.join(HERE, "lora_keys.json"), "w"), indent=1)


def loss_golden():
    from loss import ContrastiveLoss
    out = {}
    for ci, (b, d, temp) in enumerate([(4, 64, 0.07), (32, 768, 0.07), (8, 32, 1e-6), (8, 32, 1e6)]):
        torch.manual_seed(200 + ci)
        m = ContrastiveLoss(init_temp=temp)
        t = torch.randn(b, d, requires_grad=True)
        i = torch.randn(b, d, requires_grad=True)
        l = m(t, i)
        l.backward()
        out.update({f"c{ci}_t": t, f"c{ci}_i": i, f"c{ci}_log_temp": m.log_temp.detach().clone(), f"c{ci}_loss": l,
                    f"c{ci}_dt": t.grad, f"c{ci}_di": i.grad, f"c{ci}_dlog_temp": m.log_temp.grad})
    npz("contrastive.npz", **out)


def text_encoder_golden():
    from encoder.text_encoder import TextEncoder
    torch.manual_seed(300)
    m = TextEncoder(vocab_size=200, embed_dim=32, num_layers=2, n_head=4, projection_dim=24, dropout=0.2)
    ids = torch.randint(4, 200, (3, 9))
    mask = torch.ones(3, 9, dtype=torch.long)
    mask[0, 6:] = 0
    mask[2, 3:] = 0
    ids = ids * mask
    out = {"ids": ids, "mask": mask}
    for k, v in m.state_dict().items():
        out["sd__" + k] = v
    m.eval()
    with torch.no_grad():
        out["eval_seq"] = m(ids, mask, return_pooled=False)
        out["eval_pooled"] = m(ids, mask, return_pooled=True)
    # train-mode math with dropout disabled (every row computed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for layer in m.transformer.layers:
        layer.self_attn.dropout = 0.0
    m.train()
    out["train_seq"] = m(ids, mask, return_pooled=False).detach()
    out["train_pooled"] = m(ids, mask, return_pooled=True).detach()
    npz("text_encoder.npz", **out)
    # default configuration: shapes only
    d = TextEncoder()
    json.dump({"default_param_count": sum(p.numel() for p in d.parameters()),
               "default_keys": sorted(d.state_dict().keys())[:12],
               "n_keys": len(d.state_dict())}, open(os.path.join(HERE, "text_encoder_meta.json"), "w"), indent=1)


def encoder_step_golden():
    """One contrastive pre-training step of the reference (encoder/train_encoder.py:103-108: TextEncoder in training
    mode -> pooled embeddings -> ContrastiveLoss against image embeddings -> backward), dropout disabled so that the
    arithmetic is reproducible: inputs, weights, loss and the gradient of every parameter."""
    from encoder.text_encoder import TextEncoder
    from loss import ContrastiveLoss
    torch.manual_seed(400)
    m = TextEncoder(vocab_size=300, embed_dim=96, num_layers=2, n_head=4, projection_dim=64, dropout=0.2)
    crit = ContrastiveLoss(0.07)
    B, T = 8, 12
    lens = torch.tensor([12, 5, 9, 1, 12, 7, 3, 10])
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()
    ids = torch.randint(4, 300, (B, T)) * mask
    ids[:, 0] = 1
    img = torch.randn(B, 64, requires_grad=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for layer in m.transformer.layers:
        layer.self_attn.dropout = 0.0
    m.train()
    pooled = m(input_ids=ids, attention_mask=mask)
    loss = crit(pooled, img)
    loss.backward()
    out = {"ids": ids, "mask": mask, "img": img.detach(), "pooled": pooled.detach(), "loss": loss.detach(),
           "d_img": img.grad, "d_log_temp": crit.log_temp.grad, "log_temp": crit.log_temp.detach()}
    for k, v in m.state_dict().items():
        out["sd__" + k] = v
    for k, v in m.named_parameters():
        out["grad__" + k] = v.grad
    npz("encoder_step.npz", **out)


D96 = dict(vocab_size=300, embed_dim=192, num_layers=2, n_head=2, projection_dim=64, dropout=0.2)


def text_encoder_d96_golden():
    """The reference TextEncoder at its own head width (encoder/text_encoder.py:6-27: 768 / 8 = 96) on a narrower
    model: 192 wide, 2 heads of 96.  Weights are NOT stored: tests/seeded_weights.py regenerates them.  Stored: eval-mode
    sequence and pooled outputs, train-mode sequence output (dropout disabled), and one contrastive step
    (encoder/train_encoder.py:103-108) with the loss and the gradient of every parameter."""
    sys.path.insert(0, os.path.dirname(HERE))
    from seeded_weights import text_encoder_state
    from encoder.text_encoder import TextEncoder
    from loss import ContrastiveLoss
    torch.manual_seed(500)
    m = TextEncoder(**D96)
    sd = text_encoder_state(500, D96["vocab_size"], D96["embed_dim"], D96["num_layers"], D96["projection_dim"])
    assert sorted(sd) == sorted(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    B, T = 8, 12
    lens = torch.tensor([12, 5, 9, 1, 12, 7, 3, 10])
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()
    ids = torch.randint(4, 300, (B, T)) * mask
    ids[:, 0] = 1
    out = {"ids": ids, "mask": mask}
    m.eval()
    with torch.no_grad():
        out["eval_seq"] = m(ids, mask, return_pooled=False)
        out["eval_pooled"] = m(ids, mask, return_pooled=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for layer in m.transformer.layers:
        layer.self_attn.dropout = 0.0
    m.train()
    out["train_seq"] = m(ids, mask, return_pooled=False).detach()
    crit = ContrastiveLoss(0.07)
    img = torch.randn(B, D96["projection_dim"], requires_grad=True)
    pooled = m(input_ids=ids, attention_mask=mask)
    loss = crit(pooled, img)
    loss.backward()
    out.update({"img": img.detach(), "pooled": pooled.detach(), "loss": loss.detach(), "d_img": img.grad,
                "d_log_temp": crit.log_temp.grad, "log_temp": crit.log_temp.detach()})
    for k, v in m.named_parameters():
        out["grad__" + k] = v.grad
    npz("text_encoder_d96.npz", **out)


def collate_golden():
    from utils.collate_fn import skip_broken_collate_fn
    torch.manual_seed(400)
    a = {"image": torch.randn(3, 4, 4), "text": "a b"}
    b = {"image": torch.randn(3, 4, 4), "text": "c"}
    bad = {"image": torch.full((3, 4, 4), float("nan")), "text": "d"}
    r1 = skip_broken_collate_fn([a, None, b])
    r2 = skip_broken_collate_fn([None, None])
    r3 = skip_broken_collate_fn([a, bad])
    npz("collate.npz", a=a["image"], b=b["image"], r1_images=r1["image"])
    json.dump({"r1_texts": r1["text"], "r1_keys": sorted(r1.keys()), "r2_is_none": r2 is None, "r3_is_none": r3 is None},
              open(os.path.join(HERE, "collate.json"), "w"))


def tokenizer_golden():
    import tempfile
    from tokenizer.train_tokenizer import train_tokenizer
    from tokenizer.tokenizer import load_tokenizer
    tags = ["1girl blue_archive shiroko_(blue_archive) solo", "1boy red_hair smile", "landscape mountain sky cloud",
            "1girl long_hair looking_at_viewer smile", "cat sitting window sunlight", "2girls school_uniform outdoors",
            "no_humans scenery tree river", "1girl blue_eyes white_dress beach"] * 4
    with tempfile.TemporaryDirectory() as td:
        tags_path = os.path.join(td, "tags.txt")
        open(tags_path, "w").write("\n".join(tags) + "\n")
        cwd = os.getcwd()
        os.chdir(td)
        try:
            train_tokenizer(tags_path, vocab_size=300, output_dir=td)
        finally:
            os.chdir(cwd)
        tj = open(os.path.join(td, "tokenizer.json")).read()
        open(os.path.join(HERE, "tokenizer_tiny.json"), "w").write(tj)
        tok = load_tokenizer(os.path.join(td, "tokenizer.json"))
        texts = ["1girl blue_archive shiroko_(blue_archive)", "cat sitting", "", "unknownword sky " * 40]
        enc77 = tok(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt")
        encl = tok(texts[:3], padding=True, max_length=128, truncation=True, return_tensors="pt")
        json.dump({"texts": texts, "ids77": enc77.input_ids.tolist(), "mask77": enc77.attention_mask.tolist(),
                   "ids_longest": encl.input_ids.tolist(), "mask_longest": encl.attention_mask.tolist(),
                   "pad": tok.pad_token_id, "cls": tok.cls_token_id, "sep": tok.sep_token_id},
                  open(os.path.join(HERE, "tokenizer_cases.json"), "w"))


def torch_host_logic_golden():
    from torch.utils.data import DistributedSampler
    cases = []
    ds = list(range(103))
    for world in (1, 2, 4, 8):
        for epoch in (1, 2):
            for drop_last in (False, True):
                for rank in range(world):
                    s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, drop_last=drop_last)
                    s.set_epoch(epoch)
                    cases.append({"n": 103, "world": world, "rank": rank, "epoch": epoch, "drop_last": drop_last,
                                  "idx": list(iter(s))})
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-6)
    lrs = []
    for e in [1.0, 1.25, 5.5, 9.99, 10.0, 10.5, 29.9, 30.0, 31.7, 70.0, 100.3]:
        sch.step(e)
        lrs.append([e, opt.param_groups[0]["lr"]])
    json.dump({"sampler": cases, "lr": lrs}, open(os.path.join(HERE, "host_logic.json"), "w"))


if __name__ == "__main__":
    lora_golden()
    loss_golden()
    text_encoder_golden()
    encoder_step_golden()
    text_encoder_d96_golden()
    collate_golden()
    try:
        tokenizer_golden()
    except Exception as e:  # tokenizer training needs `tokenizers`; record why if it fails
        print("tokenizer golden failed:", repr(e))
    torch_host_logic_golden()
    print("done")
