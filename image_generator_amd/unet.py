"""The text-conditioned denoiser on gfx950 kernels.

Same call surface as the network the reference instantiates at
/root/reference/generator/image_generator.py:27-29 and calls at :93-95 / :124:
`unet(sample, timestep, encoder_hidden_states=...).sample`, `unet.config.in_channels`,
`.parameters()`, `.state_dict()` (parameter names follow the public SD-v1 checkpoint layout).

Design (MI355X-first, not a module-by-module translation):
  * activations are bf16 NHWC / token-major end to end - GroupNorm, convolution and attention all
    see C-contiguous rows and the (B,C,H,W)<->(B,HW,C) reshapes of the transformer wrapper are free;
  * parameters are fp32 masters living in ONE flat buffer; a single cast kernel refreshes the flat
    bf16 operand copy; because q/k/v (and cross-attention k/v) weights are adjacent in that buffer the
    three projections run as one GEMM on a [3C][C] view;
  * conv weights are stored [Cout][kh][kw][Cin] (torch channels_last for the OIHW parameter) which is
    the K-contiguous operand of the implicit GEMM; the data-gradient operands (transposed,
    tap-flipped) are refreshed once per optimizer step;
  * bias, timestep-embedding broadcast and residual adds are fused into the GEMM epilogues.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Sequence

import torch
import torch.nn as nn

from ._lib import lib, check
from . import ops
from .ops import (BF16, F32, AttentionFn, CatNormShortcutFn, ContextKVFn, ConvFn, FeedForwardFn, GegluFn, GegluLinearFn, GroupNormFn, LayerNormFn, LinearFn,
                  SiluF32ToBf16Fn, SplitColsFn, ToNCHWFn, ToNHWCFn, WeightPack, _stream)


@dataclass
class DenoiserConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
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
        return DenoiserConfig(block_out_channels=(64, 128, 128), layers_per_block=1,
                              down_has_attn=(True, True, False), num_heads=4, cross_attention_dim=64,
                              norm_num_groups=8, sample_size=8)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


# the first convolution's per-sample bias gradient comes out of the second normalisation's backward kernel
_COLSUM_FUSION = os.environ.get("MI355_NO_COLSUM_FUSION") != "1"


# data gradients read the weights as stored instead of per-step transposed copies
_BT_DGRAD = os.environ.get("MI355_NO_BT_DGRAD") != "1"

# no-gradient calls: the transformer blocks' LayerNorms folded into the products around them
_LN_FOLD = os.environ.get("MI355_NO_LN_FOLD") != "1"


def _ceil8(n):
    return (n + 7) // 8 * 8


def _bt_ok(n_in, n_out, weights):
    """Whether a layer's data gradient reads its weight as stored (ops.gemm_bt / conv2d_dgrad_bt): the layer's inputs in
    whole 160-column tiles, its outputs in whole 64-deep k-tiles.  Trainable weights only - a frozen layer's transposed
    copy is built once, and the product on it is ~5 % faster.  Not the 320-channel layers: their large maps run on tile
    shapes that have no such operand mode, and their weights are the small ones."""
    return (_BT_DGRAD and n_in % 160 == 0 and n_out % 64 == 0 and min(n_in, n_out) >= 640
            and all(w.requires_grad for w in weights))


# ------------------------------------------------------------------------------------------------
# leaf layers
# ------------------------------------------------------------------------------------------------
class HipLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)
        self.pack = WeightPack()
        self.pack.n, self.pack.k = out_features, in_features

    def forward(self, x, residual=None, out_fp32=False):
        return LinearFn.apply(x, self.pack, self.bias, residual, out_fp32, self.weight)


class HipConv2d(nn.Module):
    """weight is an OIHW parameter whose memory is [O][kh][kw][I] (channels_last)."""

    def __init__(self, cin, cout, ksize, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = cin, cout, ksize, stride
        w = torch.empty(cout, cin, ksize, ksize)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        bound = 1 / math.sqrt(cin * ksize * ksize)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        p = self.pack = WeightPack()
        p.ksize, p.cin, p.cout, p.cin_pad, p.cout_pad = ksize, cin, cout, _ceil8(cin), _ceil8(cout)
        p.n, p.k = p.cout_pad, ksize * ksize * p.cin_pad

    @property
    def padded(self):
        return self.pack.cin != self.pack.cin_pad or self.pack.cout != self.pack.cout_pad

    def forward(self, x, rowbias=None, residual=None, ups=False, out_fp32=False):
        if self.kernel_size == 1 and self.stride == 1 and not self.padded:
            y = LinearFn.apply(x, self.pack, self.bias, residual, out_fp32, self.weight)
        else:
            y = ConvFn.apply(x, self.weight, self.bias, rowbias, residual, self.pack, self.stride, ups, out_fp32)
        if self.pack.gn_unit:           # the statistics of the GroupNorm(s) that read this output came out of its epilogue
            pre = ops.take_gn_pre()
            if pre is not None:
                y._gn_pre = pre
        return y


class HipGroupNorm(nn.Module):
    def __init__(self, groups, channels, eps):
        super().__init__()
        self.num_groups, self.eps = groups, eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.grad_dst = None        # (dgamma, dbeta) views of the flat gradient buffer in direct-gradient mode

    def forward(self, x, silu, passthrough=False, colsum=False):
        """passthrough: returns (y, x') where x' is x for the branch that bypasses the norm (see GroupNormFn).
        colsum: x comes from a convolution with a per-sample bias, whose gradient the backward kernel then also produces."""
        pre = getattr(x, "_gn_pre", None)
        if passthrough and not (torch.is_grad_enabled() and x.requires_grad):
            return GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, silu, self.grad_dst, False, False, pre), x
        return GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, silu, self.grad_dst, passthrough, colsum, pre)


class HipLayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.grad_dst = None        # [2][C] view (dgamma | dbeta) of the flat gradient buffer in direct-gradient mode

    def forward(self, x, passthrough=False):
        if passthrough and not (torch.is_grad_enabled() and x.requires_grad):
            return LayerNormFn.apply(x, self.weight, self.bias, self.eps, self.grad_dst), x
        return LayerNormFn.apply(x, self.weight, self.bias, self.eps, self.grad_dst, passthrough)


# ------------------------------------------------------------------------------------------------
# blocks
# ------------------------------------------------------------------------------------------------
class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear_1 = HipLinear(in_dim, out_dim)
        self.linear_2 = HipLinear(out_dim, out_dim)

    def forward(self, t_emb):
        h = self.linear_1(t_emb, out_fp32=True)
        return self.linear_2(SiluF32ToBf16Fn.apply(h), out_fp32=True)


class ResnetBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, eps)
        self.conv1 = HipConv2d(cin, cout, 3)
        self.time_emb_proj = HipLinear(temb_dim, cout)
        self.norm2 = HipGroupNorm(groups, cout, eps)
        self.conv2 = HipConv2d(cout, cout, 3)
        self.conv_shortcut = HipConv2d(cin, cout, 1) if cin != cout else None
        self._tproj = None          # this block's slice of the model-wide fused time-embedding projection

    def forward(self, x, temb_act):
        h, x = self.norm1(x, silu=True, passthrough=True)      # x: the shortcut branch's view of the input
        rb = self._tproj if self._tproj is not None else self.time_emb_proj(temb_act, out_fp32=True)
        h = self.conv1(h, rowbias=rb)
        h = self.norm2(h, silu=True, colsum=_COLSUM_FUSION and rb.requires_grad)
        res = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return self.conv2(h, residual=res)

    def forward_cat(self, x1, x2, temb_act):
        """Inference only: the block on torch.cat([x1, x2], -1) (hidden state | skip connection) without building the
        concatenation - the first normalisation reads both tensors, the 1x1 shortcut is two products that add up."""
        sc = self.conv_shortcut
        h = ops.groupnorm_cat(x1, x2, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, True)
        rb = self._tproj if self._tproj is not None else self.time_emb_proj(temb_act, out_fp32=True)
        h = self.conv1(h, rowbias=rb)
        h = self.norm2(h, silu=True, colsum=_COLSUM_FUSION and rb.requires_grad)
        B, H, W, C1 = x1.shape
        C2 = x2.shape[-1]
        w = sc.pack.w16.view(sc.pack.cout, C1 + C2)
        if C1 % 64 == 0 and C2 % 64 == 0 and sc.pack.cout % 160 == 0:
            res = ops.gemm_cat(x1.view(-1, C1), x2.view(-1, C2), w, bias=sc.bias)       # one product, both sources in place
        else:
            res = ops.gemm(x1.view(-1, C1), w[:, :C1], bias=sc.bias)
            res = ops.gemm(x2.view(-1, C2), w[:, C1:], residual=res, out=res)
        return self.conv2(h, residual=res.view(B, H, W, -1))

    def forward_cat_train(self, x1, x2, temb_act):
        """The same with autograd: the two input gradients leave as separate tensors (ops.CatNormShortcutFn)."""
        n1, sc = self.norm1, self.conv_shortcut
        h, res = CatNormShortcutFn.apply(x1, x2, n1.weight, n1.bias, n1.num_groups, n1.eps, True, n1.grad_dst, sc.pack,
                                         sc.bias, sc.weight, getattr(x1, "_gn_pre", None), getattr(x2, "_gn_pre", None))
        rb = self._tproj if self._tproj is not None else self.time_emb_proj(temb_act, out_fp32=True)
        h = self.conv1(h, rowbias=rb)
        h = self.norm2(h, silu=True, colsum=_COLSUM_FUSION and rb.requires_grad)
        return self.conv2(h, residual=res)

    def cat_train_ready(self, x1, x2):
        sc = self.conv_shortcut
        return (torch.is_grad_enabled() and sc is not None and sc.kernel_size == 1 and not sc.padded
                and x1.shape[-1] % 160 == 0 and x2.shape[-1] % 8 == 0 and x1.is_contiguous() and x2.is_contiguous()
                and x1.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
                and ops.cat_train_ready(sc.pack, sc.weight) and os.environ.get("MI355_NO_CAT_FUSION") != "1")

    def cat_ready(self, x1, x2):
        sc = self.conv_shortcut
        return (not torch.is_grad_enabled() and sc is not None and sc.kernel_size == 1 and not sc.padded
                and x1.shape[-1] % 8 == 0 and x2.shape[-1] % 8 == 0 and x1.is_contiguous() and x2.is_contiguous()
                and os.environ.get("MI355_NO_CAT_FUSION") != "1")


class context_kv_cache:
    """`with context_kv_cache():` - inside, under no_grad, every cross-attention layer projects its text states to keys
    and values ONCE and re-uses them: the text states of a sampling loop are the same in all of its steps
    (image_generator.py:76-78,93-95), so 16 GEMMs per step become 16 per image batch.  The caller promises that
    `encoder_hidden_states` and the weights do not change inside the block."""
    active = None

    def __enter__(self):
        self._prev = context_kv_cache.active
        context_kv_cache.active = {} if os.environ.get("MI355_NO_KV_CACHE") != "1" else None
        return self

    def __exit__(self, *exc):
        context_kv_cache.active = self._prev
        return False


class Attention(nn.Module):
    def __init__(self, query_dim, heads, context_dim=None):
        super().__init__()
        self.heads = heads
        self.is_cross = context_dim is not None
        context_dim = context_dim or query_dim
        self.to_q = HipLinear(query_dim, query_dim, bias=False)
        self.to_k = HipLinear(context_dim, query_dim, bias=False)
        self.to_v = HipLinear(context_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([HipLinear(query_dim, query_dim), nn.Identity()])
        # fused projection operand (row-stacked adjacent weights), filled in by the owning UNet
        self.fused_pack = WeightPack()
        if self.is_cross:
            self.fused_pack.n, self.fused_pack.k = 2 * query_dim, context_dim
        else:
            self.fused_pack.n, self.fused_pack.k = 3 * query_dim, query_dim
        self.lora_fused = None      # stacked views of the projections' adapters (set by the owning UNet)
        self._ctx_kv = None         # this layer's slice of the model-wide text key / value projection (one forward pass)

    def _plain(self):
        return all(isinstance(m, HipLinear) for m in (self.to_q, self.to_k, self.to_v))

    def context_kv(self, context):
        """The text keys | values [B, T, 2C] of this (plain, fused) cross-attention layer: the sampling loop's cached
        projection, this forward pass's slice of the model-wide product, or a product of its own."""
        cache = context_kv_cache.active if not torch.is_grad_enabled() else None
        kv = None if cache is None else cache.get(id(self))
        if kv is not None and (kv[0] is not context or kv[1].shape[:2] != context.shape[:2]):
            kv = None
        if kv is None and self._ctx_kv is not None:
            kv = (context, self._ctx_kv)
            if cache is not None:
                cache[id(self)] = kv
        if kv is None:
            kv = (context, LinearFn.apply(context, self.fused_pack, None, None, False, self.to_k.weight,
                                          self.to_v.weight))
            if cache is not None:
                cache[id(self)] = kv
        return kv[1]

    def forward(self, x, context, residual):
        if self._plain() and self.fused_pack.w16 is not None:
            if self.is_cross:
                q = self.to_q(x)
                a = AttentionFn.apply(q, self.context_kv(context), None, self.heads)
            else:
                qkv = LinearFn.apply(x, self.fused_pack, None, None, False, self.to_q.weight, self.to_k.weight,
                                     self.to_v.weight)
                a = AttentionFn.apply(qkv, None, None, self.heads)
        elif self.lora_fused is not None and self.fused_pack.w16 is not None:
            # adapters around projections that share their input: one GEMM for the row-stacked base weights, the
            # adapters' rank-4 terms in its epilogue
            from .lora import LoraFusedFn
            lf = self.lora_fused
            if self.is_cross:
                kv = LoraFusedFn.apply_to(context, self.fused_pack, lf)
                a = AttentionFn.apply(self.to_q(x).contiguous(), kv, None, self.heads)
            else:
                a = AttentionFn.apply(LoraFusedFn.apply_to(x, self.fused_pack, lf), None, None, self.heads)
        else:   # adapters (LoRALinear) wrapped around the projections
            src = context if self.is_cross else x
            a = AttentionFn.apply(self.to_q(x).contiguous(), self.to_k(src).contiguous(), self.to_v(src).contiguous(),
                                  self.heads)
        return self.to_out[0](a, residual=residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = HipLinear(dim_in, dim_out * 2)

    def forward(self, x):
        pk = self.proj.pack
        if isinstance(self.proj, HipLinear) and pk.w16 is not None and pk.k % 64 == 0 and (pk.n // 2) % 80 == 0:
            # the gate is applied in the GEMM epilogue
            if not torch.is_grad_enabled():         # inference: the [M][8C] pre-activation never reaches HBM
                y = ops.gemm_geglu(x.reshape(-1, x.shape[-1]), pk.w16.view(pk.n, pk.k), self.proj.bias)
                return y.view(*x.shape[:-1], pk.n // 2)
            return GegluLinearFn.apply(x, pk, self.proj.bias, self.proj.weight)
        return GegluFn.apply(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), HipLinear(dim * 4, dim)])

    def forward(self, x, residual):
        g, o = self.net[0].proj, self.net[2]
        if (torch.is_grad_enabled() and isinstance(g, HipLinear) and isinstance(o, HipLinear) and g.pack.w16 is not None
                and o.pack.w16 is not None and (o.pack.w16_t is not None or o.pack.bt) and g.pack.k % 64 == 0 and (g.pack.n // 2) % 80 == 0
                and os.environ.get("MI355_NO_FF_FUSION") != "1"):
            # training: one autograd node, the gate's backward rides the output projection's data-gradient GEMM
            return FeedForwardFn.apply(x, residual, g.pack, g.bias, g.weight, o.pack, o.bias, o.weight)
        return self.net[2](self.net[0](x), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, context_dim):
        super().__init__()
        self.norm1 = HipLayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = HipLayerNorm(dim)
        self.attn2 = Attention(dim, heads, context_dim)
        self.norm3 = HipLayerNorm(dim)
        self.ff = FeedForward(dim)

    # ---- no-gradient path with the three LayerNorms folded into the products around them ------------------------
    def fold_ready(self, x):
        """Whether `forward_folded` applies: no gradients, plain projections in whole 160-column tiles, enough rows
        that no product would want a split contraction (the row statistics come out of the tile epilogue)."""
        C = x.shape[-1]
        a1, a2, g, o = self.attn1, self.attn2, self.ff.net[0].proj, self.ff.net[2]
        return (not torch.is_grad_enabled() and _LN_FOLD and C % 160 == 0 and C % 64 == 0
                and x.numel() // C >= 4096 and x.dtype == BF16
                and a1._plain() and a2._plain() and a1.fused_pack.w16 is not None and a2.fused_pack.w16 is not None
                and all(isinstance(m, HipLinear) and m.pack.w16 is not None for m in (a1.to_out[0], a2.to_out[0], a2.to_q, g, o))
                and g.pack.k % 64 == 0 and (g.pack.n // 2) % 160 == 0)

    def _folded(self):
        """The folded operands (gamma-scaled bf16 weights, their column sums, beta-shifted biases) of the three
        products that follow norm1 / norm2 / norm3, rebuilt when a weight or a norm parameter changed."""
        a1, a2, g = self.attn1, self.attn2, self.ff.net[0].proj
        src = (a1.to_q.weight, a1.to_k.weight, a1.to_v.weight, a2.to_q.weight, g.weight, g.bias, self.norm1.weight,
               self.norm1.bias, self.norm2.weight, self.norm2.bias, self.norm3.weight, self.norm3.bias)
        sig = tuple(p._version for p in src) + (src[0].data_ptr(),)
        if getattr(self, "_fold_sig", None) != sig:
            C, dev = a1.to_q.weight.shape[1], a1.to_q.weight.device
            wp = torch.empty((3 * C, C), dtype=BF16, device=dev)
            cs, bo = torch.empty(3 * C, dtype=F32, device=dev), torch.empty(3 * C, dtype=F32, device=dev)
            # the query projections also carry the softmax scale and the change of base (ops.attention_fwd, prescaled)
            qs = (C // a1.heads) ** -0.5 * ops.ATTN_PRESCALE
            for i, m in enumerate((a1.to_q, a1.to_k, a1.to_v)):
                sl = slice(i * C, (i + 1) * C)
                ops.ln_fold_weight(m.weight.detach(), self.norm1.weight.detach(), self.norm1.bias.detach(), None,
                                   out=(wp[sl], cs[sl], bo[sl]), scale=qs if i == 0 else 1.0)
            q = ops.ln_fold_weight(a2.to_q.weight.detach(), self.norm2.weight.detach(), self.norm2.bias.detach(), None,
                                   scale=(C // a2.heads) ** -0.5 * ops.ATTN_PRESCALE)
            f = ops.ln_fold_weight(g.weight.detach(), self.norm3.weight.detach(), self.norm3.bias.detach(), g.bias.detach())
            self._fold, self._fold_sig = ((wp, cs, bo), q, f), sig
        return self._fold

    def forward_folded(self, x, context, stat, parts):
        """x [B, N, C] un-normalised with the row statistics its producer emitted (ops.gemm_rowstats): the block without
        a LayerNorm launch - every product that follows a norm applies it in its epilogue, every product that writes
        the residual stream emits the next norm's statistics."""
        B, N, C = x.shape
        a1, a2, o = self.attn1, self.attn2, self.ff.net[2]
        (w1, c1, b1), (w2, c2, b2), (w3, c3, b3) = self._folded()
        x0 = x.view(B * N, C)
        qkv = ops.gemm_ln(x0, w1, c1, b1, stat, parts, self.norm1.eps).view(B, N, 3 * C)
        a = ops.attention_fwd(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], a1.heads, prescaled=True)[0].view(B * N, C)
        p = a1.to_out[0].pack
        x1, st1, p1 = ops.gemm_rowstats(a, p.w16.view(p.n, p.k), bias=a1.to_out[0].bias, residual=x0)
        q = ops.gemm_ln(x1, w2, c2, b2, st1, p1, self.norm2.eps).view(B, N, C)
        kv = a2.context_kv(context)
        a = ops.attention_fwd(q, kv[:, :, :C], kv[:, :, C:], a2.heads, prescaled=True)[0].view(B * N, C)
        p = a2.to_out[0].pack
        x2, st2, p2 = ops.gemm_rowstats(a, p.w16.view(p.n, p.k), bias=a2.to_out[0].bias, residual=x1)
        h = ops.gemm_ln(x2, w3, c3, b3, st2, p2, self.norm3.eps, geglu=True)
        return ops.gemm(h, o.pack.w16.view(o.pack.n, o.pack.k), bias=o.bias, residual=x2).view(B, N, C)

    def forward(self, x, context):
        # each norm hands back the residual branch's view of its input: the two gradients meet inside the
        # normalisation's backward kernel instead of in a separate addition
        n, x = self.norm1(x, passthrough=True)
        x = self.attn1(n, None, x)
        n, x = self.norm2(x, passthrough=True)
        x = self.attn2(n, context, x)
        n, x = self.norm3(x, passthrough=True)
        return self.ff(n, x)


class Transformer2D(nn.Module):
    def __init__(self, channels, heads, context_dim, groups):
        super().__init__()
        self.norm = HipGroupNorm(groups, channels, 1e-6)
        self.proj_in = HipConv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, context_dim)])
        self.proj_out = HipConv2d(channels, channels, 1)

    def forward(self, x, context):
        B, H, W, C = x.shape
        h, x = self.norm(x, silu=False, passthrough=True)
        blk0, pi = self.transformer_blocks[0], self.proj_in
        if (len(self.transformer_blocks) == 1 and not pi.padded and pi.kernel_size == 1 and pi.pack.w16 is not None
                and blk0.fold_ready(h)):
            # sampling: proj_in emits the first norm's row statistics, the block runs without LayerNorm launches
            h, stat, parts = ops.gemm_rowstats(h.view(-1, C), pi.pack.w16.view(pi.pack.n, pi.pack.k), bias=pi.bias)
            h = blk0.forward_folded(h.view(B, H * W, C), context, stat, parts)
            return self.proj_out(h.view(B, H, W, C), residual=x)
        h = self.proj_in(h).view(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        return self.proj_out(h.view(B, H, W, C), residual=x)


class Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, stride=2)

    def forward(self, x):
        return self.conv(x)


class Upsample(nn.Module):
    """nearest 2x upsample fused into the convolution's tile loader (never materialised)."""

    def __init__(self, c):
        super().__init__()
        self.conv = HipConv2d(c, c, 3)

    def forward(self, x):
        return self.conv(x, ups=True)


class DownBlock(nn.Module):
    def __init__(self, cfg, cin, cout, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList() if has_attn else None
        for j in range(cfg.layers_per_block):
            self.resnets.append(ResnetBlock(cin if j == 0 else cout, cout, cfg.time_embed_dim, cfg.norm_num_groups,
                                            cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2D(cout, cfg.num_heads, cfg.cross_attention_dim, cfg.norm_num_groups))
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for j, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg, c):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(c, c, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2D(c, cfg.num_heads, cfg.cross_attention_dim, cfg.norm_num_groups)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
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
            self.resnets.append(ResnetBlock(rin + skip, cout, cfg.time_embed_dim, cfg.norm_num_groups, cfg.norm_eps))
            if has_attn:
                self.attentions.append(Transformer2D(cout, cfg.num_heads, cfg.cross_attention_dim, cfg.norm_num_groups))
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for j, r in enumerate(self.resnets):
            skip = skips.pop()
            if r.cat_ready(x, skip):
                x = r.forward_cat(x, skip, temb)
            elif r.cat_train_ready(x, skip):
                x = r.forward_cat_train(x, skip, temb)
            else:
                x = r(torch.cat([x, skip], dim=-1), temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: DenoiserConfig | None = None):
        super().__init__()
        cfg = cfg or DenoiserConfig.sd_v1()
        self.config = cfg
        ch = list(cfg.block_out_channels)
        self.conv_in = HipConv2d(cfg.in_channels, ch[0], 3)
        self.time_embedding = TimestepEmbedding(ch[0], cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownBlock(cfg, cin, cout, cfg.down_has_attn[i], i != len(ch) - 1))
        self.mid_block = MidBlock(cfg, ch[-1])
        self.up_blocks = nn.ModuleList()
        rev, rev_attn = ch[::-1], list(cfg.down_has_attn)[::-1]
        cout = rev[0]
        for i, c in enumerate(rev):
            cprev, cout = cout, c
            cin = rev[min(i + 1, len(ch) - 1)]
            self.up_blocks.append(UpBlock(cfg, cin, cout, cprev, rev_attn[i], i != len(ch) - 1))
        self.conv_norm_out = HipGroupNorm(cfg.norm_num_groups, ch[0], cfg.norm_eps)
        self.conv_out = HipConv2d(ch[0], cfg.out_channels, 3)
        # every layer whose output is read by a GroupNorm emits that norm's statistics from its own epilogue, in runs of
        # channels that divide every group that can contain them (the up path concatenates maps of different widths:
        # groups of 30 or 60 channels over sources whose own groups are 10, 20 or 40 wide)
        unit = 10 if all(c % 320 == 0 for c in ch) and cfg.norm_num_groups == 32 else (8 if all((c // cfg.norm_num_groups) % 8 == 0 for c in ch) else 0)
        if unit and os.environ.get("MI355_NO_GN_PRE") != "1":
            for m in self.modules():
                if isinstance(m, ResnetBlock):
                    m.conv1.pack.gn_unit = m.conv2.pack.gn_unit = unit
                elif isinstance(m, Transformer2D):
                    m.proj_out.pack.gn_unit = unit
                elif isinstance(m, (Downsample, Upsample)):
                    m.conv.pack.gn_unit = unit
            self.conv_in.pack.gn_unit = unit
        # flat-buffer state (built lazily on the device the parameters live on)
        self._flat = None
        self._flat16 = None
        self._flat_grad = None
        self._frozen_base = False
        self._layout = None
        self._sig = None

    # -------------------------------------------------------------------------------------------
    # flat parameter storage and bf16 operand copies
    # -------------------------------------------------------------------------------------------
    def _gemm_weight_ids(self):
        ids = set()
        for m in self.modules():
            if isinstance(m, (HipLinear, HipConv2d)):
                ids.add(id(m.weight))
        return ids

    def _build_flat(self):
        params = list(self.parameters())
        self._plist = params                    # cached: walking the module tree on every forward costs milliseconds
        self._resnets = [m for m in self.modules() if isinstance(m, ResnetBlock)]
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("UNet2DConditionModel: parameters must live on the HIP device (no CPU path)")
        wid = self._gemm_weight_ids()
        # the time-embedding projections of all residual blocks read the same input: their weights (and their
        # biases) are laid out back to back so that ONE GEMM computes all of them
        self._temb_layers = [m.time_emb_proj for m in self.modules()
                             if isinstance(m, ResnetBlock) and isinstance(m.time_emb_proj, HipLinear)]
        tw = {id(l.weight) for l in self._temb_layers}
        tb = {id(l.bias) for l in self._temb_layers}
        from .lora import LoRALinear
        self._adapters = [m for m in self.modules() if isinstance(m, LoRALinear)]
        ad = {id(p) for m in self._adapters for p in (m.lora_A.weight, m.lora_B.weight)}
        # adapters around projections that share an input are laid out [A_0 .. A_S-1 | B_0 .. B_S-1] so that their
        # stacked factors are views of the flat storage
        fused_groups, ordered, seen = [], [], set()
        for m in self.modules():
            if isinstance(m, Attention):
                m.lora_fused = None
                ws = [m.to_k, m.to_v] if m.is_cross else [m.to_q, m.to_k, m.to_v]
                if (all(isinstance(w, LoRALinear) and isinstance(w.original, HipLinear) and w.r == 4 for w in ws)
                        and len({w.alpha for w in ws}) == 1):
                    fused_groups.append((m, ws))
                    for w in ws:
                        ordered.append(w.lora_A.weight)
                    for w in ws:
                        ordered.append(w.lora_B.weight)
                    seen.update(id(p) for w in ws for p in (w.lora_A.weight, w.lora_B.weight))
        big = [l.weight for l in self._temb_layers] + [p for p in params if id(p) in wid and id(p) not in tw]
        small = [l.bias for l in self._temb_layers] + [p for p in params if id(p) not in wid and id(p) not in tb and id(p) not in ad]
        # last, contiguous: the trainable range of a LoRA fine-tune
        adapters = ordered + [p for p in params if id(p) in ad and id(p) not in seen]
        offs, off = {}, 0
        for p in big + small + adapters:
            offs[id(p)] = off
            off += _ceil8(p.numel())
        self._n_big = sum(_ceil8(p.numel()) for p in big)
        self._adapter_range = (off - sum(_ceil8(p.numel()) for p in adapters), off)
        # operand copies depend on the GEMM weights (and on the biases of channel-padded convs) only
        self._sig_params = [p for p in params if id(p) in wid] + [m.bias for m in self.modules()
                                                                  if isinstance(m, HipConv2d) and m.padded]
        flat = torch.zeros(off, dtype=F32, device=dev)
        for p in big + small + adapters:
            o, n = offs[id(p)], p.numel()
            if p.dim() == 4:
                co, ci, kh, kw = p.shape
                view = flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                view = flat[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._flat, self._offs = flat, offs
        self._flat16 = torch.zeros(off, dtype=BF16, device=dev)
        self._flat_grad = None
        self._zero_table = None
        # operand packs
        self._packs = []
        lora_wrapped = {id(m.original) for m in self.modules() if isinstance(m, LoRALinear)}
        for m in self.modules():
            if isinstance(m, HipLinear):
                pk, o = m.pack, offs[id(m.weight)]
                pk.w16 = self._flat16[o:o + pk.n * pk.k].view(pk.n, pk.k)
                pk.bt = _bt_ok(pk.k, pk.n, [m.weight]) and id(m) not in lora_wrapped
                pk.w16_t = None if pk.bt else torch.empty((pk.k, pk.n), dtype=BF16, device=dev)
                self._packs.append(("linear", m, pk, o))
            elif isinstance(m, HipConv2d):
                pk, o = m.pack, offs[id(m.weight)]
                t = pk.ksize * pk.ksize
                if m.padded:
                    pk.w16 = torch.zeros((pk.cout_pad, t, pk.cin_pad), dtype=BF16, device=dev)
                    pk.bias_pad = torch.zeros(pk.cout_pad, dtype=F32, device=dev)
                else:
                    pk.w16 = self._flat16[o:o + pk.cout * t * pk.cin].view(pk.cout, t, pk.cin)
                pk.bt = not m.padded and _bt_ok(pk.cin, pk.cout, [m.weight])
                pk.w16_t = None if pk.bt else torch.empty((pk.cin_pad, t, pk.cout_pad), dtype=BF16, device=dev)
                self._packs.append(("conv", m, pk, o))
        for m in self.modules():
            if isinstance(m, Attention):
                pk = m.fused_pack
                ws = [m.to_k, m.to_v] if m.is_cross else [m.to_q, m.to_k, m.to_v]
                ws = [w.original if isinstance(w, LoRALinear) else w for w in ws]
                if not all(isinstance(w, HipLinear) for w in ws):
                    pk.w16 = None
                    continue
                o0 = offs[id(ws[0].weight)]
                adjacent = all(offs[id(ws[i + 1].weight)] == offs[id(ws[i].weight)] + ws[i].weight.numel()
                               for i in range(len(ws) - 1))
                if not adjacent:
                    pk.w16 = None
                    continue
                pk.w16 = self._flat16[o0:o0 + pk.n * pk.k].view(pk.n, pk.k)
                pk.bt = _bt_ok(pk.k, pk.n, [w.weight for w in ws]) and not any(id(w) in lora_wrapped for w in ws)
                pk.w16_t = None if pk.bt else torch.empty((pk.k, pk.n), dtype=BF16, device=dev)
                self._packs.append(("fused", m, pk, o0))
        from types import SimpleNamespace
        for m, ws in fused_groups:
            if m.fused_pack.w16 is None:
                continue
            S, r, K, C = len(ws), ws[0].r, ws[0].lora_A.weight.shape[1], ws[0].lora_B.weight.shape[0]
            oa, ob = offs[id(ws[0].lora_A.weight)], offs[id(ws[0].lora_B.weight)]
            m.lora_fused = SimpleNamespace(
                S=S, r=r, alpha=float(ws[0].alpha), mods=ws, grad_dst=None,
                A_all=flat[oa:oa + S * r * K].view(S * r, K), B_all=flat[ob:ob + S * C * r].view(S * C, r),
                params=[p for w in ws for p in (w.lora_A.weight, w.lora_B.weight)])
        # the text keys / values of every cross-attention layer multiply the same text states: one packed bf16 operand
        # [sum 2 C_i][K] (refreshed with the other operand copies) lets ONE product compute all of them
        self._ctx_layers, self._ctx_w16, self._ctx_table = [], None, None
        cl = [m for m in self.modules() if isinstance(m, Attention) and m.is_cross and m._plain() and m.fused_pack.w16 is not None]
        if (len(cl) > 1 and len({m.fused_pack.k for m in cl}) == 1 and all(m.fused_pack.n % 8 == 0 for m in cl)
                and cl[0].fused_pack.k % 8 == 0 and os.environ.get("MI355_NO_CTX_KV_FUSION") != "1"):
            K = cl[0].fused_pack.k
            self._ctx_layers = cl
            self._ctx_w16 = torch.zeros((sum(m.fused_pack.n for m in cl), K), dtype=BF16, device=dev)
            rows, r0 = [], 0
            for m in cl:
                pk = m.fused_pack
                rows.append([pk.w16.data_ptr(), self._ctx_w16.data_ptr() + 2 * r0 * K, 2 * pk.n * K])
                r0 += pk.n
            assert all(v % 16 == 0 for r in rows for v in r)
            self._ctx_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._temb_pack = None
        tl = self._temb_layers
        if len(tl) > 1 and all(l.weight.numel() % 8 == 0 and l.bias.numel() % 8 == 0 for l in tl):
            pk = self._temb_pack = WeightPack()
            pk.n, pk.k = sum(l.out_features for l in tl), tl[0].in_features
            o0, b0 = offs[id(tl[0].weight)], offs[id(tl[0].bias)]
            pk.w16 = self._flat16[o0:o0 + pk.n * pk.k].view(pk.n, pk.k)
            pk.w16_t = torch.empty((pk.k, pk.n), dtype=BF16, device=dev)
            pk.bias_parts = len(tl)
            self._temb_bias = flat[b0:b0 + pk.n]
            self._temb_offs = (o0, b0)
            self._temb_sizes = tuple(l.out_features for l in tl)
            members = {id(l) for l in tl}          # their own packs are never used: no per-layer transposes
            self._packs = [e for e in self._packs if not (e[0] == "linear" and id(e[1]) in members)]
            self._packs.append(("temb", None, pk, o0))
        # one job table for every per-step weight transpose: (src, dst, R, C, T, first tile)
        jobs, first = [], 0
        for kind, m, pk, o in self._packs:
            if pk.bt:
                continue
            if kind in ("linear", "fused", "temb"):
                R, C, T = pk.n, pk.k, 1
            else:
                R, C, T = pk.cout_pad, pk.cin_pad, pk.ksize * pk.ksize
            jobs.append([pk.w16.data_ptr(), pk.w16_t.data_ptr(), R, C, T, first])
            first += ((R + 63) // 64) * ((C + 63) // 64) * T
        self._tjobs = torch.tensor(jobs, dtype=torch.int64, device=dev)
        self._ttiles = first
        self._layout = (dev, params[0].data_ptr())
        self._sig = None

    def _ensure_flat(self):
        p0 = next(self.parameters())
        if self._flat is None or self._layout != (p0.device, p0.data_ptr()):
            self._build_flat()

    def _signature(self):
        return tuple(p._version for p in self._sig_params)

    def refresh_operands(self, cast=True):
        """Re-derive every bf16 operand from the fp32 masters (call after the parameters changed)."""
        self._ensure_flat()
        s = _stream()
        if cast:
            check(lib.mi355_cast_f32_to_bf16(self._flat.data_ptr(), self._flat16.data_ptr(), self._flat.numel(), s), "cast")
        for kind, m, pk, o in self._packs:
            if kind == "conv" and m.padded:
                t = pk.ksize * pk.ksize
                phys = m.weight.data.permute(0, 2, 3, 1).reshape(pk.cout, t, pk.cin)
                pk.w16[:pk.cout, :, :pk.cin] = phys.to(BF16)
                pk.bias_pad[:pk.cout] = m.bias.data
        if self._tjobs.shape[0]:
            check(lib.mi355_transpose_batched(self._tjobs.data_ptr(), self._tjobs.shape[0], self._ttiles, s), "transpose_batched")
        if self._ctx_table is not None:
            check(lib.mi355_copy_ranges(self._ctx_table.data_ptr(), self._ctx_table.shape[0], 64, s), "copy_ranges")
        self._sig = self._signature()

    def _maybe_refresh(self):
        self._ensure_flat()
        if self._sig != self._signature():
            self.refresh_operands()

    def operands_current(self):
        """Tell the module its bf16 copies already match the masters (fused optimizer wrote both)."""
        self._sig = self._signature()

    def flat_parameters(self):
        self._ensure_flat()
        return self._flat, self._flat16

    def trainable_range(self):
        """[start, end) of the flat storage the optimizer owns: everything, or - with a frozen base and LoRA adapters
        as the only trainable parameters - the adapters' contiguous tail."""
        self._ensure_flat()
        req = [p.requires_grad for p in self._plist]
        if all(req):
            return 0, self._flat.numel()
        ad = {id(p) for m in self._adapters for p in (m.lora_A.weight, m.lora_B.weight)}
        if ad and all(p.requires_grad == (id(p) in ad) for p in self._plist):
            return self._adapter_range
        raise NotImplementedError("fused training supports all parameters trainable, or exactly the LoRA adapters")

    def enable_direct_grads(self):
        """Weight gradients of all GEMM layers are written straight into one flat fp32 buffer
        (`param.grad` are views of it); small parameters accumulate into its tail through autograd.
        With a frozen base the buffer covers the adapters only and their kernels accumulate into it."""
        self._ensure_flat()
        t0, t1 = self.trainable_range()
        self._frozen_base = t0 != 0
        if self._flat_grad is None or self._flat_grad.numel() != t1 - t0:
            self._flat_grad = torch.zeros(t1 - t0, dtype=F32, device=self._flat.device)
            self._zero_table = None
            self._grad_pad = None
        g = self._flat_grad
        if self._frozen_base:
            for p in self._plist:
                p.grad = None
            for kind, m, pk, o in self._packs:
                pk.grad_out = pk.bias_grad_out = None
            for m in self.modules():
                if isinstance(m, (HipGroupNorm, HipLayerNorm)):
                    m.grad_dst = None
            # The adapter gradients come out of the grouped weight-gradient launch zero-padded to 8 ranks
            # (dA_pad [8][K], dB_pad [N][8]); `finish_grads()` compacts them into the flat buffer with one launch.
            if any(m.r > 8 for m in self._adapters):
                raise NotImplementedError("fused LoRA training supports ranks up to 8")
            groups = [m.lora_fused for m in self.modules() if isinstance(m, Attention) and m.lora_fused is not None]
            in_group = {id(w) for lf in groups for w in lf.mods}
            singles = [m for m in self._adapters if id(m) not in in_group]

            def width(rt):
                return 8 if rt <= 8 else 16
            n_pad = sum(8 * (m.lora_A.weight.shape[1] + m.lora_B.weight.shape[0]) for m in singles)
            n_pad += sum(width(lf.S * lf.r) * (lf.A_all.shape[1] + lf.B_all.shape[0]) for lf in groups)
            pad = self._grad_pad = torch.zeros(n_pad, dtype=F32, device=g.device)
            op16 = self._lora_op16 = torch.zeros(n_pad, dtype=BF16, device=g.device)    # A16 [W][K] | Bt16 [W][N] per entry
            table, ptable, q = [], [], 0
            for m in self._adapters:
                A, Bm = m.lora_A.weight, m.lora_B.weight
                r, K, N = A.shape[0], A.shape[1], Bm.shape[0]
                oa, ob = self._offs[id(A)] - t0, self._offs[id(Bm)] - t0
                A.grad, Bm.grad = g[oa:oa + r * K].view(r, K), g[ob:ob + N * r].view(N, r)
                m.grad_dst = None
            for m in singles:
                A, Bm = m.lora_A.weight, m.lora_B.weight
                r, K, N = A.shape[0], A.shape[1], Bm.shape[0]
                ga, gb = pad[q:q + 8 * K].view(8, K), pad[q + 8 * K:q + 8 * (K + N)].view(N, 8)
                a16, bt16 = op16[q:q + 8 * K].view(8, K), op16[q + 8 * K:q + 8 * (K + N)].view(8, N)
                q += 8 * (K + N)
                m.grad_dst = (ga, gb, a16, bt16)
                ptable.append([0, A.data_ptr(), a16.data_ptr(), r * K, 0, 0])
                ptable.append([1, Bm.data_ptr(), bt16.data_ptr(), N, r, N])
                table.append([ga.data_ptr(), A.grad.data_ptr(), r, K, K, K])
                table.append([gb.data_ptr(), Bm.grad.data_ptr(), N, r, 8, r])
            for lf in groups:
                S, r, K, N = lf.S, lf.r, lf.A_all.shape[1], lf.B_all.shape[0]
                W, C = width(S * r), N // S
                ga, gb = pad[q:q + W * K].view(W, K), pad[q + W * K:q + W * (K + N)].view(N, W)
                a16, bt16 = op16[q:q + W * K].view(W, K), op16[q + W * K:q + W * (K + N)].view(W, N)
                q += W * (K + N)
                lf.grad_dst = (ga, gb, a16, bt16)
                ptable.append([0, lf.A_all.data_ptr(), a16.data_ptr(), S * r * K, 0, 0])
                for i, w in enumerate(lf.mods):      # adapter i: rows i*r .. i*r + r, columns i*C .. i*C + C
                    ptable.append([1, w.lora_B.weight.data_ptr(), bt16.data_ptr() + 2 * (i * r * N + i * C), C, r, N])
                # the group's A gradients are adjacent in the flat buffer (same order as the factors themselves)
                table.append([ga.data_ptr(), lf.mods[0].lora_A.weight.grad.data_ptr(), S * r, K, K, K])
                for i, w in enumerate(lf.mods):
                    table.append([gb.data_ptr() + 4 * (i * C * W + i * r), w.lora_B.weight.grad.data_ptr(), C, r, W, r])
            self._compact_table = torch.tensor(table, dtype=torch.int64, device=g.device)
            self._pack_table = torch.tensor(ptable, dtype=torch.int64, device=g.device)
            self._grad_packs = []
            return g
        for p in self.parameters():
            o, n = self._offs[id(p)], p.numel()
            if p.dim() == 4:
                co, ci, kh, kw = p.shape
                p.grad = g[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                p.grad = g[o:o + n].view(p.shape)
        for m in self._adapters:
            m.grad_dst = None
        for m in self.modules():
            if isinstance(m, Attention) and m.lora_fused is not None:
                m.lora_fused.grad_dst = None
        fused_members = set()
        for kind, m, pk, o in self._packs:
            if kind == "fused":
                pk.grad_out = g[o:o + pk.n * pk.k].view(pk.n, pk.k)
                for w in ([m.to_k, m.to_v] if m.is_cross else [m.to_q, m.to_k, m.to_v]):
                    fused_members.add(id(w))
            elif kind == "temb":
                pk.bias_grad_out = g[self._temb_offs[1]:self._temb_offs[1] + pk.n]
                for l in self._temb_layers:
                    fused_members.add(id(l))
        self._grad_packs = []          # (offset, numel, pack) of every layer whose backward writes into `g`
        for kind, m, pk, o in self._packs:
            if kind == "conv" and m.padded:
                continue
            pk.grad_out = g[o:o + pk.n * pk.k].view(pk.n, pk.k)
            if kind in ("fused", "temb") or id(m) not in fused_members:
                self._grad_packs.append((o, pk.n * pk.k, pk))
            if kind not in ("fused", "temb") and m.bias is not None:
                ob = self._offs[id(m.bias)]
                pk.bias_grad_out = g[ob:ob + m.bias.numel()]
        for m in self.modules():
            if isinstance(m, HipGroupNorm):
                ow, ob, c = self._offs[id(m.weight)], self._offs[id(m.bias)], m.weight.numel()
                m.grad_dst = (g[ow:ow + c], g[ob:ob + c])
            elif isinstance(m, HipLayerNorm):
                ow, ob, c = self._offs[id(m.weight)], self._offs[id(m.bias)], m.weight.numel()
                m.grad_dst = g[ow:ow + 2 * c].view(2, c) if ob == ow + c else None
        return g

    def finish_grads(self):
        """After backward (and after the weight-gradient queue was joined): compact the rank-padded adapter gradients
        into the flat buffer (frozen-base mode only)."""
        if self._frozen_base and self._flat_grad is not None:
            t = self._compact_table
            check(lib.mi355_copy2d_f32_batched(t.data_ptr(), t.shape[0], _stream()), "copy2d_batched")

    def zero_small_grads(self, weights_too=True):
        """Direct-gradient mode: clear what the backward pass ACCUMULATES into.  Bias sums, small parameters and
        the weights that go through autograd's `+=` (channel-padded convs) always; the big weight region only when
        `weights_too` - a weight-gradient queue with `zero_split_outputs` clears its pixel-split destinations
        itself, and unsplit jobs overwrite theirs, so ~95 % of the 3.4 GB buffer need not be touched."""
        if self._flat_grad is None:
            return
        if self._frozen_base:       # an adapter that does not run backward this step must end up with a zero gradient
            check(lib.mi355_zero_f32(self._grad_pad.data_ptr(), self._grad_pad.numel(), _stream()), "zero")
            # this step's bf16 operands of the adapters' skinny GEMMs (the factors changed in the last optimizer step)
            t = self._pack_table
            check(lib.mi355_lora_pack(t.data_ptr(), t.shape[0], _stream()), "lora_pack")
            return
        if weights_too:
            check(lib.mi355_zero_f32(self._flat_grad.data_ptr(), self._flat_grad.numel(), _stream()), "zero")
            return
        if self._zero_table is None:
            g = self._flat_grad
            ranges = [(g.data_ptr() + 4 * self._n_big, 4 * (g.numel() - self._n_big))]
            for kind, m, pk, o in self._packs:
                if kind == "conv" and m.padded:
                    ranges.append((g.data_ptr() + 4 * o, 4 * _ceil8(m.weight.numel())))
            assert all(a % 16 == 0 and b % 16 == 0 for a, b in ranges)
            self._zero_table = torch.tensor([v for r in ranges for v in r], dtype=torch.int64, device=g.device)
        check(lib.mi355_zero_ranges(self._zero_table.data_ptr(), self._zero_table.numel() // 2, _stream()), "zero_ranges")

    # -------------------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states=None):
        cfg = self.config
        div = 1 << (len(cfg.block_out_channels) - 1)
        if sample.dim() != 4 or sample.shape[1] != cfg.in_channels or sample.shape[2] % div or sample.shape[3] % div:
            # every level halves the map and the skip connections are concatenated without resampling
            raise ValueError(f"UNet2DConditionModel: expected [B, {cfg.in_channels}, H, W] with H and W multiples of {div}, "
                             f"got {tuple(sample.shape)}")
        if encoder_hidden_states is None or encoder_hidden_states.shape[-1] != cfg.cross_attention_dim:
            raise ValueError(f"UNet2DConditionModel: encoder_hidden_states [B, T, {cfg.cross_attention_dim}] required")
        if not sample.is_cuda:
            raise RuntimeError("UNet2DConditionModel: HIP device tensors required (no CPU path)")
        self._maybe_refresh()
        B = sample.shape[0]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        t = timestep.to(device=sample.device, dtype=F32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        t_emb = ops.timestep_embedding(t, cfg.block_out_channels[0])
        emb = self.time_embedding(t_emb)                 # [B, 4*C0] fp32
        temb_act = SiluF32ToBf16Fn.apply(emb)            # shared by every residual block
        resnets = self._resnets if self._temb_pack is not None else []
        if resnets:
            tl = self._temb_layers
            tproj = LinearFn.apply(temb_act, self._temb_pack, self._temb_bias, None, True,
                                   *[l.weight for l in tl], *[l.bias for l in tl])
            for r, sl in zip(resnets, SplitColsFn.apply(tproj, self._temb_sizes)):
                r._tproj = sl
        ctx = encoder_hidden_states
        if ctx.dtype != BF16:
            ctx = ctx.to(BF16)
        ctx = ctx.contiguous()
        ctx_layers = self._ctx_layers
        if ctx_layers:
            cache = context_kv_cache.active if not torch.is_grad_enabled() else None
            if cache is not None and all(id(m) in cache and cache[id(m)][0] is ctx for m in ctx_layers):
                ctx_layers = []             # a sampling loop's later steps: every layer re-uses its cached slice
            else:
                kvs = ContextKVFn.apply(ctx, self._ctx_w16, [m.fused_pack for m in ctx_layers],
                                        *[w for m in ctx_layers for w in (m.to_k.weight, m.to_v.weight)])
                for m, kv in zip(ctx_layers, kvs):
                    m._ctx_kv = kv
        x = ToNHWCFn.apply(sample, self.conv_in.pack.cin_pad)
        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb_act, ctx)
            skips.extend(outs)
        x = self.mid_block(x, temb_act, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, temb_act, ctx)
        for r in resnets:
            r._tproj = None
        for m in ctx_layers:
            m._ctx_kv = None
        x = self.conv_norm_out(x, silu=True)
        x = self.conv_out(x, out_fp32=True)
        return UNetOutput(ToNCHWFn.apply(x, cfg.out_channels))
