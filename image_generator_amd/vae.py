"""Latent auto-encoder (`self.vae` of /root/reference/generator/image_generator.py:31-32) on the HIP kernels.

Call surface used by the reference: `vae.encode(image).latent_dist.sample()` (:107-108),
`vae.decode(latents).sample` (:69), frozen (`requires_grad_(False)`, :40), scale 0.18215 applied by the
caller.  Architecture and parameter names are those of the public SD-v1 `AutoencoderKL` checkpoint
(83,653,863 parameters), so its state dict loads unchanged; nothing is downloaded here, weights are
random-initialised until `load_state_dict` is given real ones.

Inference only (the reference never trains it): NHWC bf16 activations, every convolution / projection is the
library's implicit-GEMM kernel, GroupNorm(+SiLU) its NHWC kernel; the down-samplers use the bottom/right-padded
stride-2 entry point, the up-samplers read their input through the fused nearest-2x gather.  The mid-block
attention has ONE head of dim 512 - beyond the flash kernel's 160 - and runs as QK^T GEMM -> row softmax ->
PV GEMM per sample (scores are 2 MB per sample at 256x256).  fp32 master parameters, bf16 operand copies
cached until a parameter changes.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from ._lib import lib, check
from .ops import BF16, F32, GroupNormFn, _stream


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


def _ceil8(n):
    return (n + 7) // 8 * 8


class _Conv(nn.Module):
    """OIHW fp32 parameter (checkpoint layout); operand copy [Cout_pad][k*k][Cin_pad] bf16, channels padded to 8."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        conv = nn.Conv2d(cin, cout, k)                         # default init of the reference's layer type
        self.weight, self.bias = nn.Parameter(conv.weight.data), nn.Parameter(conv.bias.data)
        self._sig = None

    def operand(self):
        sig = (self.weight._version, self.bias._version, self.weight.device)
        if self._sig != sig:
            ci, co, t = _ceil8(self.cin), _ceil8(self.cout), self.k * self.k
            w = torch.zeros((co, t, ci), dtype=BF16, device=self.weight.device)
            w[:self.cout, :, :self.cin] = self.weight.detach().permute(0, 2, 3, 1).reshape(self.cout, t, self.cin).to(BF16)
            b = torch.zeros(co, dtype=F32, device=self.weight.device)
            b[:self.cout] = self.bias.detach()
            self._w16, self._b, self._sig = w, b, sig
        return self._w16, self._b

    def forward(self, x, residual=None, ups=False, asym_down=False, out_fp32=False):
        w, b = self.operand()
        co = w.shape[0]
        if asym_down:
            B, H, W, C = x.shape
            out = torch.empty((B, H // 2, W // 2, co), dtype=BF16, device=x.device)
            ws = ops.workspace(x.device)
            check(lib.mi355_conv2d_down_asym_bf16(x.data_ptr(), B, H, W, C, w.data_ptr(), co, out.data_ptr(), b.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream()), "conv2d_down_asym")
            return out
        if self.k == 1:
            B, H, W, C = x.shape
            y = ops.gemm(x.view(-1, C), w.view(co, C), bias=b, out_fp32=out_fp32,
                         residual=None if residual is None else residual.view(-1, co))
            return y.view(B, H, W, co)
        return ops.conv2d(x, w, co, self.k, self.stride, ups, bias=b, residual=residual, out_fp32=out_fp32)


class _GroupNorm(nn.Module):
    def __init__(self, groups, c, eps):
        super().__init__()
        self.groups, self.eps = groups, eps
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))

    def forward(self, x, silu):
        return GroupNormFn.apply(x, self.weight, self.bias, self.groups, self.eps, silu, None)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = _GroupNorm(groups, cin, eps)
        self.conv1 = _Conv(cin, cout, 3)
        self.norm2 = _GroupNorm(groups, cout, eps)
        self.conv2 = _Conv(cout, cout, 3)
        self.conv_shortcut = _Conv(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(self.norm1(x, True))
        res = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return self.conv2(self.norm2(h, True), residual=res)


class _MidAttention(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.c = c
        self.group_norm = _GroupNorm(groups, c, eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])
        self._sig = None

    def _operands(self):
        ps = [self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight]
        sig = tuple(p._version for p in ps) + (ps[0].device,)
        if self._sig != sig:
            self._wqkv = torch.cat([p.detach() for p in ps[:3]], 0).to(BF16).contiguous()
            self._wo = ps[3].detach().to(BF16).contiguous()
            self._sig = sig
        bqkv = torch.cat([self.to_q.bias, self.to_k.bias, self.to_v.bias]).detach()
        return self._wqkv, bqkv, self._wo

    def forward(self, x):
        B, H, W, C = x.shape
        N = H * W
        wqkv, bqkv, wo = self._operands()
        t = self.group_norm(x, False).view(B * N, C)
        qkv = ops.gemm(t, wqkv, bias=bqkv).view(B, N, 3 * C)
        k = qkv[:, :, C:2 * C].contiguous()
        v = qkv[:, :, 2 * C:].contiguous()
        vt = torch.empty((B, C, N), dtype=BF16, device=x.device)
        check(lib.mi355_transpose_bf16(v.data_ptr(), vt.data_ptr(), B, N, C, _stream()), "transpose")
        o = torch.empty((B, N, C), dtype=BF16, device=x.device)
        probs = torch.empty((N, N), dtype=BF16, device=x.device)
        for b in range(B):
            s = ops.gemm(qkv[b, :, :C], k[b])                                  # [N, N] = q_b k_b^T
            check(lib.mi355_softmax_rows_bf16(s.data_ptr(), probs.data_ptr(), N, N, float(C) ** -0.5, _stream()), "softmax")
            ops.gemm(probs, vt[b], out=o[b])                                   # [N, C] = P v_b
        y = ops.gemm(o.view(B * N, C), wo, bias=self.to_out[0].bias, residual=x.view(B * N, C))
        return y.view(B, H, W, C)


class _MidBlock(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([_MidAttention(c, groups, eps)])
        self.resnets = nn.ModuleList([_Resnet(c, c, groups, eps), _Resnet(c, c, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Sampler(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = _Conv(c, c, 3, stride)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, n, down, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups, eps) for i in range(n)])
        self.downsamplers = nn.ModuleList([_Sampler(cout, 2)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(x, asym_down=True)
        return x


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups, eps) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Sampler(cout, 1)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(x, ups=True)
        return x


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, e = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = _Conv(cfg.in_channels, ch[0], 3)
        self.down_blocks = nn.ModuleList([_DownBlock(ch[max(i - 1, 0)], ch[i], cfg.layers_per_block, i < len(ch) - 1, g, e)
                                          for i in range(len(ch))])
        self.mid_block = _MidBlock(ch[-1], g, e)
        self.conv_norm_out = _GroupNorm(g, ch[-1], e)
        self.conv_out = _Conv(ch[-1], 2 * cfg.latent_channels, 3)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(self.conv_norm_out(self.mid_block(x), True))


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, e = tuple(reversed(cfg.block_out_channels)), cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = _Conv(cfg.latent_channels, ch[0], 3)
        self.mid_block = _MidBlock(ch[0], g, e)
        self.up_blocks = nn.ModuleList([_UpBlock(ch[max(i - 1, 0)], ch[i], cfg.layers_per_block + 1, i < len(ch) - 1, g, e)
                                        for i in range(len(ch))])
        self.conv_norm_out = _GroupNorm(g, ch[-1], e)
        self.conv_out = _Conv(ch[-1], cfg.out_channels, 3)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_norm_out(x, True), out_fp32=True)


class DiagonalGaussianDistribution:
    """mean / clamped log-variance of the encoder moments; `sample()` draws with torch's generator, as the
    reference's `latent_dist.sample()` does (a few thousand elements per image)."""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, device=self.mean.device,
                                                  dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, config: VaeConfig | None = None):
        super().__init__()
        self.config = config or VaeConfig.sd_v1()
        c = self.config
        self.encoder, self.decoder = _Encoder(c), _Decoder(c)
        self.quant_conv = _Conv(2 * c.latent_channels, 2 * c.latent_channels, 1)
        self.post_quant_conv = _Conv(c.latent_channels, c.latent_channels, 1)
        self.requires_grad_(False)

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL: HIP device tensors required (no CPU path)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("AutoencoderKL: inference only (the reference keeps it frozen); call under "
                               "torch.no_grad() or leave requires_grad off")

    def encode(self, image):
        """image (B,3,H,W) in [-1,1], H and W multiples of 2^(levels-1) -> .latent_dist over (B,4,H/8,W/8)."""
        self._check(image)
        with torch.no_grad():
            x = ops.nchw_to_nhwc_bf16(image, _ceil8(self.config.in_channels))
            m = self.quant_conv(self.encoder(x), out_fp32=True)                 # (B,h,w,8) fp32
            moments = ops.ToNCHWFn.apply(m.contiguous(), 2 * self.config.latent_channels)
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, latents):
        self._check(latents)
        with torch.no_grad():
            z = ops.nchw_to_nhwc_bf16(latents, _ceil8(self.config.latent_channels))
            y = self.decoder(self.post_quant_conv(z))                           # (B,H,W,8) fp32
            img = ops.ToNCHWFn.apply(y.contiguous(), self.config.out_channels)
        return SimpleNamespace(sample=img)
