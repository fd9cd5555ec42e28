"""CPU restatements of the reference's own small modules (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Each function cites the reference lines it follows.  These restatements are pinned by
golden vectors minted from the reference modules themselves (tests/golden/make_golden.py
imports /root/reference/{lora,loss}.py, encoder/text_encoder.py, utils/collate_fn.py in the
build container and stores inputs + outputs under tests/golden/).

The scheduler functions restate diffusers==0.33.1 DDPMScheduler/DDIMScheduler with the
`CompVis/stable-diffusion-v1-4/scheduler` configuration (third-party, not vendored in the
reference, not installed here): PARITY UNPINNED for those two; the reference's call sites are
/root/reference/generator/image_generator.py:86-99,114-123.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# LoRA  (/root/reference/lora.py:23-24)
# ----------------------------------------------------------------------------------------------
def lora_linear(x, w, b, lora_a, lora_b, alpha):
    """y = x W^T + b + alpha * ((x A^T) B^T)."""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return y + alpha * ((x @ lora_a.t()) @ lora_b.t())


def lora_init_bound(in_features: int) -> float:
    """kaiming_uniform_(a=sqrt(5)) bound on an (r, in) matrix: 1/sqrt(in)  (lora.py:16)."""
    gain = math.sqrt(2.0 / (1 + 5.0))
    return gain * math.sqrt(3.0 / in_features)


# ----------------------------------------------------------------------------------------------
# Contrastive loss  (/root/reference/loss.py:12-28)
# ----------------------------------------------------------------------------------------------
def contrastive_loss(text, image, log_temp):
    t = text / text.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    i = image / image.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    temp = log_temp.exp().clamp(1e-4, 1e4)
    logits = (t @ i.t()) / temp
    n = logits.shape[0]
    idx = torch.arange(n)
    lse_r = torch.logsumexp(logits, dim=1)
    lse_c = torch.logsumexp(logits, dim=0)
    diag = logits[idx, idx]
    return 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())


# ----------------------------------------------------------------------------------------------
# MSE objective (/root/reference/generator/image_generator.py:125)
# ----------------------------------------------------------------------------------------------
def mse_loss(pred, target):
    return ((pred.float() - target.float()) ** 2).mean()


# ----------------------------------------------------------------------------------------------
# Text encoder (/root/reference/encoder/text_encoder.py:16-43)
# ----------------------------------------------------------------------------------------------
def text_encoder_forward(sd: dict, input_ids, attention_mask, n_head: int, return_pooled: bool,
                         zero_padded_rows: bool):
    """Explicit math of embedding -> N x post-LN encoder layer (GELU) -> LN -> Linear.

    `sd` is the module's state dict (names of torch.nn.TransformerEncoder).
    `zero_padded_rows=True` reproduces the inference fast path the reference runs under
    no_grad/eval (image_generator.py:56): rows of padded tokens leave the transformer stack as
    zeros; with False every row is computed as in training mode with dropout disabled.
    """
    x = sd["embedding.weight"][input_ids]                                     # (B,T,D)
    b, t, d = x.shape
    pad = ~attention_mask.bool() if attention_mask is not None else torch.zeros(b, t, dtype=torch.bool)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.layers."))
    hd = d // n_head
    for li in range(n_layers):
        p = f"transformer.layers.{li}."
        qkv = x @ sd[p + "self_attn.in_proj_weight"].t() + sd[p + "self_attn.in_proj_bias"]
        q, k, v = qkv.split(d, dim=-1)
        q = q.view(b, t, n_head, hd).transpose(1, 2)
        k = k.view(b, t, n_head, hd).transpose(1, 2)
        v = v.view(b, t, n_head, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(b, t, d)
        a = a @ sd[p + "self_attn.out_proj.weight"].t() + sd[p + "self_attn.out_proj.bias"]
        x = F.layer_norm(x + a, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        f = F.gelu(x @ sd[p + "linear1.weight"].t() + sd[p + "linear1.bias"])
        f = f @ sd[p + "linear2.weight"].t() + sd[p + "linear2.bias"]
        x = F.layer_norm(x + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    if zero_padded_rows:
        x = x.masked_fill(pad[:, :, None], 0.0)
    x = F.layer_norm(x, (d,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    if return_pooled:
        y = x[:, 0] @ sd["projection.weight"].t() + sd["projection.bias"]
        return y / y.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return x @ sd["projection.weight"].t() + sd["projection.bias"]


# ----------------------------------------------------------------------------------------------
# Collate (/root/reference/utils/collate_fn.py:4-17)
# ----------------------------------------------------------------------------------------------
def skip_broken_collate(batch):
    batch = [b for b in batch if b is not None]
    if not batch:
        return None
    images = torch.stack([b["image"] for b in batch], dim=0)
    texts = [b["text"] for b in batch]
    if torch.isnan(images).any():
        return None
    return {"image": images, "text": texts}


# ----------------------------------------------------------------------------------------------
# Schedulers (third-party spec; see module docstring)
# ----------------------------------------------------------------------------------------------
def sd_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_add_noise(x0, noise, t, acp):
    a = acp[t].sqrt().view(-1, 1, 1, 1)
    s = (1.0 - acp[t]).sqrt().view(-1, 1, 1, 1)
    return a * x0 + s * noise


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    ratio = num_train_timesteps // num_inference_steps
    return (np.arange(0, num_inference_steps) * ratio).round()[::-1].astype(np.int64) + steps_offset


def ddim_step(eps, t, x, acp, num_inference_steps, num_train_timesteps=1000):
    """eta = 0, epsilon prediction, clip_sample False, set_alpha_to_one False."""
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else acp[0]
    x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
    return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps


def cfg_combine(eps_uncond, eps_text, g):
    """/root/reference/generator/image_generator.py:97-98."""
    return eps_uncond + g * (eps_text - eps_uncond)


# ----------------------------------------------------------------------------------------------
# Trainer-side host logic (/root/reference/trainer.py:105-106,261 and :108-121,230)
# ----------------------------------------------------------------------------------------------
def cosine_warm_restarts_lr(base_lr, epoch_float, t0=10, t_mult=2, eta_min=1e-6):
    """Closed form of CosineAnnealingWarmRestarts.step(epoch_float)."""
    e = epoch_float
    if t_mult == 1:
        t_i, t_cur = t0, e % t0
    elif e >= t0:
        n = int(math.log(e / t0 * (t_mult - 1) + 1, t_mult))
        t_cur = e - t0 * (t_mult ** n - 1) / (t_mult - 1)
        t_i = t0 * t_mult ** n
    else:
        t_i, t_cur = t0, e
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


def distributed_sampler_indices(n, world, rank, epoch, seed=0, drop_last=False):
    """Index list of torch.utils.data.DistributedSampler(shuffle=True) after set_epoch(epoch)."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    if drop_last and n % world != 0:
        num = math.ceil((n - world) / world)
    else:
        num = math.ceil(n / world)
    total = num * world
    if not drop_last:
        pad = total - len(idx)
        if pad <= len(idx):
            idx += idx[:pad]
        else:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world]


def lmdb_keys(idx: int):
    """/root/reference/dataset.py:215-216 and utils/make_lmdb.py:63-67."""
    return f"image-{idx:08d}".encode(), f"meta-{idx:08d}".encode()


def lmdb_length(entries: int) -> int:
    """/root/reference/dataset.py:183."""
    return entries // 2
