"""TEST INFRASTRUCTURE ONLY - CPU fp32 restatement of the latent auto-encoder the reference calls as `self.vae`
(/root/reference/generator/image_generator.py:31-32 `AutoencoderKL.from_pretrained(..., subfolder="vae")`,
used at :69 `vae.decode(latents / 0.18215).sample` and :107-109 `vae.encode(image).latent_dist.sample() * 0.18215`).

The arithmetic lives in the third-party package diffusers==0.33.1 (reference requirements.txt:8), which is neither
vendored under /root/reference nor installable here, and the reference holds no test vectors for it:
PARITY UNPINNED against that package.  This file restates the published SD-v1 `AutoencoderKL` architecture
(block_out_channels 128/256/512/512, 2 layers per block, 32 groups, eps 1e-6, single-head mid-block attention,
down-sampling convs padded on the bottom/right only, nearest-2x up-sampling) with the public checkpoint's parameter
names; it is pinned only by the 83,653,863 parameter count of that checkpoint and by internal identities.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VaeConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @staticmethod
    def sd_v1():
        return VaeConfig()

    @staticmethod
    def small():
        return VaeConfig(block_out_channels=(32, 32, 64, 64), layers_per_block=1, norm_num_groups=8)


class Resnet(nn.Module):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class MidAttention(nn.Module):
    """Single head over all H*W positions, head dim = channels."""

    def __init__(self, c, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(b, c, h, w)


class MidBlock(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([MidAttention(c, groups, eps)])
        self.resnets = nn.ModuleList([Resnet(c, c, groups, eps), Resnet(c, c, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, n, down, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups, eps) for i in range(n)])
        self.downsamplers = None
        if down:
            self.downsamplers = nn.ModuleList([nn.Module()])
            self.downsamplers[0].conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))       # zero padding on the right / bottom only
        return x


class UpBlock(nn.Module):
    def __init__(self, cin, cout, n, up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups, eps) for i in range(n)])
        self.upsamplers = None
        if up:
            self.upsamplers = nn.ModuleList([nn.Module()])
            self.upsamplers[0].conv = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class Encoder(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        ch, g, e = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([DownBlock(ch[max(i - 1, 0)], ch[i], cfg.layers_per_block, i < len(ch) - 1, g, e)
                                          for i in range(len(ch))])
        self.mid_block = MidBlock(ch[-1], g, e)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=e)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        ch, g, e = tuple(reversed(cfg.block_out_channels)), cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = MidBlock(ch[0], g, e)
        self.up_blocks = nn.ModuleList([UpBlock(ch[max(i - 1, 0)], ch[i], cfg.layers_per_block + 1, i < len(ch) - 1, g, e)
                                        for i in range(len(ch))])
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=e)
        self.conv_out = nn.Conv2d(ch[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussian:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, device=self.mean.device,
                                                  dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKLRef(nn.Module):
    def __init__(self, cfg: VaeConfig | None = None):
        super().__init__()
        self.config = cfg or VaeConfig.sd_v1()
        c = self.config
        self.encoder, self.decoder = Encoder(c), Decoder(c)
        self.quant_conv = nn.Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(c.latent_channels, c.latent_channels, 1)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))


def count_params(m):
    return sum(p.numel() for p in m.parameters())
