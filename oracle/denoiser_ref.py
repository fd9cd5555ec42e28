"""CPU fp32 oracle of the text-conditioned denoiser (TEST INFRASTRUCTURE ONLY).

This file is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

What it restates
----------------
The reference calls a third-party network at
  /root/reference/generator/image_generator.py:27-29  (construction)
  /root/reference/generator/image_generator.py:93-95  (sampling call)
  /root/reference/generator/image_generator.py:124    (training call)
namely diffusers==0.33.1 `UNet2DConditionModel` with the configuration of
`CompVis/stable-diffusion-v1-4/unet` (requirements.txt:8 of the reference pins
the version).  That package is neither vendored in the reference nor installed
here, so the arithmetic below follows its *published* architecture:

  in/out channels 4, block_out_channels (320, 640, 1280, 1280),
  layers_per_block 2, down (CrossAttnDown x3, Down), up (Up, CrossAttnUp x3),
  8 attention heads, cross_attention_dim 768, 32 norm groups, eps 1e-5,
  SiLU, flip_sin_to_cos=True, freq_shift=0, conv projections in the
  transformer wrapper (use_linear_projection=False), GEGLU feed-forward.

PARITY UNPINNED: the reference holds no test, golden vector or fixture for this
network and the third-party implementation cannot be executed here, so this
restatement is validated only by (a) the known parameter count of the SD-v1
topology (859,520,964) and (b) internal identities checked in tests/.

Everything is written with plain torch ops (conv2d, group_norm, layer_norm,
matmul, softmax) in fp32 so that it doubles as the CPU baseline.
Parameter names follow the public diffusers checkpoint layout so that a state
dict moves between this oracle and the HIP module unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class DenoiserConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # per down block: does it carry transformer (cross-attention) layers
    down_has_attn: Sequence[bool] = (True, True, True, False)
    num_heads: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64

    @staticmethod
    def sd_v1() -> "DenoiserConfig":
        return DenoiserConfig()

    @staticmethod
    def small() -> "DenoiserConfig":
        """Small plumbing configuration (64x64 images -> 8x8 latents)."""
        return DenoiserConfig(block_out_channels=(64, 128, 128), layers_per_block=1,
                              down_has_attn=(True, True, False), num_heads=4,
                              cross_attention_dim=64, norm_num_groups=8, sample_size=8)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """Sinusoidal embedding, cos first (flip_sin_to_cos), no frequency shift."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    freqs = torch.exp(exponent)
    args = timesteps.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, out_dim)
        self.linear_2 = nn.Linear(out_dim, out_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, query_dim, heads, context_dim=None):
        super().__init__()
        context_dim = context_dim or query_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(context_dim, query_dim, bias=False)
        self.to_v = nn.Linear(context_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])

    def forward(self, x, context=None):
        context = x if context is None else context
        b, n, c = x.shape
        h = self.heads
        q = self.to_q(x).view(b, n, h, c // h).transpose(1, 2)
        k = self.to_k(context).view(b, -1, h, c // h).transpose(1, 2)
        v = self.to_v(context).view(b, -1, h, c // h).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(c // h))
        p = torch.softmax(scores, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(b, n, c)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, context_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, context_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2D(nn.Module):
    def __init__(self, channels, heads, context_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, context_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, context):
        b, c, hh, ww = x.shape
        res = x
        h = self.proj_in(self.norm(x))
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        return self.proj_out(h) + res


class Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cfg, cin, cout, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList() if has_attn else None
        for j in range(cfg.layers_per_block):
            self.resnets.append(ResnetBlock(cin if j == 0 else cout, cout, cfg.time_embed_dim,
                                            cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2D(cout, cfg.num_heads, cfg.cross_attention_dim,
                                                     cfg.norm_num_groups))
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if add_down else None

    def forward(self, x, temb, context):
        outs = []
        for j, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, context)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg, c):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps),
            ResnetBlock(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)])
        self.attentions = nn.ModuleList([Transformer2D(c, cfg.num_heads, cfg.cross_attention_dim,
                                                       cfg.norm_num_groups)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cfg, cin, cout, cprev, has_attn, add_up):
        super().__init__()
        n = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList() if has_attn else None
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = cprev if j == 0 else cout
            self.resnets.append(ResnetBlock(rin + skip, cout, cfg.time_embed_dim,
                                            cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2D(cout, cfg.num_heads, cfg.cross_attention_dim,
                                                     cfg.norm_num_groups))
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if add_up else None

    def forward(self, x, skips, temb, context):
        for j, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, context)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Out:
    def __init__(self, sample):
        self.sample = sample


class DenoiserRef(nn.Module):
    """fp32 restatement of the SD-v1 style conditional UNet (see module docstring)."""

    def __init__(self, cfg: DenoiserConfig | None = None):
        super().__init__()
        cfg = cfg or DenoiserConfig.sd_v1()
        self.config = cfg
        ch = list(cfg.block_out_channels)
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownBlock(cfg, cin, cout, cfg.down_has_attn[i], i != len(ch) - 1))
        self.mid_block = MidBlock(cfg, ch[-1])
        self.up_blocks = nn.ModuleList()
        rev = ch[::-1]
        rev_attn = list(cfg.down_has_attn)[::-1]
        cout = rev[0]
        for i, c in enumerate(rev):
            cprev, cout = cout, c
            cin = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cfg, cin, cout, cprev, rev_attn[i], i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states=None):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0]) if timestep.numel() == 1 else timestep
        temb = self.time_embedding(timestep_embedding(timestep, self.config.block_out_channels[0]))
        ctx = encoder_hidden_states
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, ctx)
            skips.extend(outs)
        x = self.mid_block(x, temb, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, ctx)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return _Out(x)


def mse_objective(noise_pred: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """The training objective of /root/reference/generator/image_generator.py:125."""
    return ((noise_pred.float() - noise.float()) ** 2).mean()


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
