"""Text encoder that produces the cross-attention states of the denoiser.

Interface and parameter names of /root/reference/encoder/text_encoder.py:5-43
(`TextEncoder(vocab_size, embed_dim, num_layers, n_head, projection_dim, dropout)`,
`forward(input_ids, attention_mask, return_pooled=True)`; state-dict keys of
`nn.Embedding` + `nn.TransformerEncoder` + `nn.LayerNorm` + `nn.Linear`), so a checkpoint of the
reference module loads unchanged.

The generator uses it frozen under no_grad (image_generator.py:56-60 semantics): post-LN encoder
layers with GELU, key-padding mask, and - as the reference's inference path does - the rows of padded
tokens leave the layer stack as zeros before the final LayerNorm/projection.  In training mode
(encoder pre-training, outside the denoiser hot path) every row is computed and dropout applies.
The arithmetic here is issued through PyTorch-ROCm tensor ops on the HIP device; moving it onto the
library's LayerNorm/GEMM/attention kernels is the next widening step of this path.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _SelfAttn(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.embed_dim, self.num_heads = d, h
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayer(nn.Module):
    def __init__(self, d, h, ff, p):
        super().__init__()
        self.self_attn = _SelfAttn(d, h)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.p = p

    def forward(self, x, pad):
        b, t, d = x.shape
        h = self.self_attn.num_heads
        qkv = F.linear(x, self.self_attn.in_proj_weight, self.self_attn.in_proj_bias)
        q, k, v = (z.view(b, t, h, d // h).transpose(1, 2) for z in qkv.split(d, dim=-1))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(d // h)
        if pad is not None:
            s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        a = F.dropout(torch.softmax(s, dim=-1), self.p, self.training) @ v
        a = self.self_attn.out_proj(a.transpose(1, 2).reshape(b, t, d))
        x = self.norm1(x + F.dropout(a, self.p, self.training))
        f = self.linear2(F.dropout(F.gelu(self.linear1(x)), self.p, self.training))
        return self.norm2(x + F.dropout(f, self.p, self.training))


class _Encoder(nn.Module):
    def __init__(self, d, h, ff, p, n):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(d, h, ff, p) for _ in range(n)])


class TextEncoder(nn.Module):
    def __init__(self, vocab_size=16384, embed_dim=768, num_layers=8, n_head=8, projection_dim=768, dropout=0.2):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.transformer = _Encoder(embed_dim, n_head, embed_dim * 4, dropout, num_layers)
        self.norm = nn.LayerNorm(embed_dim)
        self.projection = nn.Linear(embed_dim, projection_dim)

    def forward(self, input_ids, attention_mask, return_pooled=True):
        x = self.embedding(input_ids)
        pad = ~attention_mask.bool() if attention_mask is not None else None
        for layer in self.transformer.layers:
            x = layer(x, pad)
        if pad is not None and not (self.training and torch.is_grad_enabled()):
            x = x.masked_fill(pad[:, :, None], 0.0)     # inference path: padded rows are zero-filled
        x = self.norm(x)
        if return_pooled:
            return F.normalize(self.projection(x[:, 0]), dim=-1)
        return self.projection(x)
