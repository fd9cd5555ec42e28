"""Text encoder that produces the cross-attention states of the denoiser.

Interface and parameter names of /root/reference/encoder/text_encoder.py:5-43
(`TextEncoder(vocab_size, embed_dim, num_layers, n_head, projection_dim, dropout)`,
`forward(input_ids, attention_mask, return_pooled=True)`; state-dict keys of
`nn.Embedding` + `nn.TransformerEncoder` + `nn.LayerNorm` + `nn.Linear`), so a checkpoint of the
reference module loads unchanged.

The generator uses it frozen under no_grad (image_generator.py:56-60 semantics): post-LN encoder
layers with GELU, key-padding mask, and - as the reference's inference path does - the rows of padded
tokens leave the layer stack as zeros before the final LayerNorm/projection.  That inference path
(`return_pooled=False`, no gradients, HIP device) runs on the library's kernels: fused in-projection GEMM,
flash attention with the key-padding mask as per-sample key counts, out-projection / feed-forward GEMMs
with the residual in the epilogue, LayerNorm and GELU kernels; bf16 activations, fp32 accumulation.
Training mode (the contrastive pre-training of encoder/train_encoder.py:83-110) and the pooled output run on the same
kernels through autograd Functions: every row is computed, as the reference's training path does, the key-padding mask
and the dropout on the attention probabilities live inside the flash-attention kernels (forward and backward), the other
three dropouts of a layer are fused with the residual add / the GELU, and the pooled output is L2-normalised by a row
kernel.  Dropout masks are hashes of (seed, position): they are regenerated in the backward pass, never stored; the
seed of a forward pass is drawn from torch's CPU generator, so `torch.manual_seed` makes a run repeatable (the masks are
not torch's own: the two random streams differ, the statistics do not).

There is one backend.  CPU tensors, and geometries the kernels do not carry (width or head width not a multiple of 8,
heads wider than 96), raise: nothing here computes the encoder with tensor ops instead.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _kernels():
    from . import ops
    from ._lib import lib, check
    return ops, lib, check


class _ToBf16Fn(torch.autograd.Function):
    """fp32 -> bf16 (the embedding rows on their way into the first GEMM); gradient back as fp32."""

    @staticmethod
    def forward(ctx, x):
        ops, lib, check = _kernels()
        return ops.cast_bf16(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        ops, lib, check = _kernels()
        dy = dy.contiguous()
        dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        check(lib.mi355_cast_bf16_to_f32(dy.data_ptr(), dx.data_ptr(), dy.numel(), ops._stream()), "cast")
        return dx


class _AttnTrainFn(torch.autograd.Function):
    """Self-attention on the fused q/k/v projection output [B, T, 3D] with the key-padding mask (valid key counts) and
    dropout on the attention probabilities, both inside the flash kernels."""

    @staticmethod
    def forward(ctx, qkv, kv_len, heads, p, seed):
        ops, lib, check = _kernels()
        B, T, C3 = qkv.shape
        C = C3 // 3
        d = C // heads
        out = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        check(lib.mi355_attention_fwd_train(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                            None if kv_len is None else kv_len.data_ptr(), p, seed, B, heads, T, T, d,
                                            C3, C3, C3, C, d ** -0.5, ops._stream()), "attention_fwd_train")
        ctx.save_for_backward(qkv, out, lse, kv_len)
        ctx.meta = (heads, p, seed)
        return out

    @staticmethod
    def backward(ctx, do):
        ops, lib, check = _kernels()
        qkv, out, lse, kv_len = ctx.saved_tensors
        heads, p, seed = ctx.meta
        B, T, C3 = qkv.shape
        C = C3 // 3
        d = C // heads
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        dq, dk, dv = dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:]
        check(lib.mi355_attention_bwd_train(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), do.data_ptr(),
                                            lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                            None if kv_len is None else kv_len.data_ptr(), p, seed, B, heads, T, T, d,
                                            C3, C3, C3, C, d ** -0.5, ops._stream()), "attention_bwd_train")
        return dqkv, None, None, None, None


class _DropoutAddFn(torch.autograd.Function):
    """y = residual + dropout_p(x): the dropout behind the attention / feed-forward output projection and the residual
    add in one pass."""

    @staticmethod
    def forward(ctx, x, residual, p, seed):
        ops, lib, check = _kernels()
        x, residual = x.contiguous(), residual.contiguous()
        y = torch.empty_like(x)
        check(lib.mi355_dropout_add_bf16(x.data_ptr(), residual.data_ptr(), y.data_ptr(), x.numel(), p, seed, ops._stream()),
              "dropout_add")
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops, lib, check = _kernels()
        p, seed = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.mi355_dropout_add_bf16(dy.data_ptr(), None, dx.data_ptr(), dy.numel(), p, seed, ops._stream()), "dropout_add")
        return dx, dy, None, None


class _GeluDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ops, lib, check = _kernels()
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.mi355_gelu_dropout_fwd(x.data_ptr(), y.data_ptr(), x.numel(), p, seed, ops._stream()), "gelu_dropout_fwd")
        ctx.save_for_backward(x)
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops, lib, check = _kernels()
        (x,) = ctx.saved_tensors
        p, seed = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib.mi355_gelu_dropout_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), p, seed, ops._stream()),
              "gelu_dropout_bwd")
        return dx, None, None


class _L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) on fp32 rows (the pooled output)."""

    @staticmethod
    def forward(ctx, x):
        ops, lib, check = _kernels()
        x = x.contiguous()
        B, D = x.shape
        y, nrm = torch.empty_like(x), torch.empty(B, dtype=torch.float32, device=x.device)
        check(lib.mi355_l2norm_rows_f32(x.data_ptr(), y.data_ptr(), nrm.data_ptr(), B, D, ops._stream()), "l2norm_rows")
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops, lib, check = _kernels()
        y, nrm = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(lib.mi355_l2norm_rows_bwd_f32(y.data_ptr(), nrm.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.shape[0], y.shape[1],
                                            ops._stream()), "l2norm_rows_bwd")
        return dx


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
    """Parameter container with `nn.TransformerEncoderLayer`'s names; the arithmetic is TextEncoder's."""

    def __init__(self, d, h, ff, p):
        super().__init__()
        self.self_attn = _SelfAttn(d, h)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.p = p


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

    # ---- inference on the HIP kernels -----------------------------------------------------------
    def _forward_hip(self, input_ids, attention_mask):
        from . import ops
        w16 = {n: pk.w16 for n, pk in self._train_packs().items()}     # one set of bf16 operand copies for both paths
        B, T = input_ids.shape
        D = self.embedding.embedding_dim
        H = self.transformer.layers[0].self_attn.num_heads
        x = ops.cast_bf16(self.embedding(input_ids).contiguous()).view(B * T, D)      # gather, then fp32 -> bf16
        kv_len, bad = self._key_counts(attention_mask)
        for i, layer in enumerate(self.transformer.layers):
            pre = f"transformer.layers.{i}."
            sa = layer.self_attn
            qkv = ops.gemm(x, w16[pre + "self_attn.in_proj_weight"], bias=sa.in_proj_bias).view(B, T, 3 * D)
            a, _ = ops.attention_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, kv_len=kv_len)
            x = ops.gemm(a.view(B * T, D), w16[pre + "self_attn.out_proj.weight"], bias=sa.out_proj.bias, residual=x)
            x = ops.layernorm_bf16(x, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            f = ops.gelu_bf16(ops.gemm(x, w16[pre + "linear1.weight"], bias=layer.linear1.bias))
            x = ops.gemm(f, w16[pre + "linear2.weight"], bias=layer.linear2.bias, residual=x)
            x = ops.layernorm_bf16(x, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        if attention_mask is not None:
            x = x.view(B, T, D).masked_fill(~attention_mask.bool()[:, :, None], 0.0).view(B * T, D)   # padded rows leave as zeros
        x = ops.layernorm_bf16(x.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps)
        y = ops.gemm(x, w16["projection.weight"], bias=self.projection.bias, out_fp32=True)
        return self._poison(y.view(B, T, -1), bad)

    # The kernels take the key-padding mask as a count of leading valid keys per sample: tokenizers pad on the right
    # (tokenizer/tokenizer.py:4-9 sets no padding side, the library default is right).  The reference honours any mask
    # (encoder/text_encoder.py:31-34), so a sample whose mask is NOT of that form - left-padded, with holes, or without
    # a single valid token - must not come back as some other arithmetic's result: the layout is verified on the device
    # (a few [B, T] tensor ops, no host synchronisation, legal inside a stream capture) and every output row of such a
    # sample is returned as NaN, which the step's non-finite guards then see.  `check_mask_layout = True` verifies on
    # the host instead (one synchronisation per call) and raises.
    check_mask_layout = False

    def _key_counts(self, attention_mask):
        """(valid-key counts [B] int32 clamped to >= 1, [B] bool: the mask of this sample is not a right-padded one)."""
        if attention_mask is None:
            return None, None
        m = attention_mask.bool()
        kv_len = m.sum(dim=1).to(torch.int32)
        T = m.shape[1]
        bad = (m != (torch.arange(T, device=m.device)[None, :] < kv_len[:, None])).any(dim=1) | (kv_len < 1)
        if self.check_mask_layout and bool(bad.any()):
            raise ValueError("TextEncoder: attention_mask must be right-padded with at least one valid token per row "
                             "(the HIP attention kernels take the mask as valid-key counts)")
        return kv_len.clamp_(min=1), bad

    @staticmethod
    def _poison(y, bad):
        if bad is None:
            return y
        return y.masked_fill(bad.view(-1, *([1] * (y.dim() - 1))), float("nan"))

    def _train_packs(self):
        """Operand packs (bf16 [N][K] and its transpose) of every weight matrix for the training path, rebuilt when a
        parameter changed; a trainer that updates the masters and the bf16 copies itself calls `operands_current()`."""
        from .ops import BF16, WeightPack, _stream
        from ._lib import lib, check
        sig = tuple(p._version for p in self.parameters()) + (next(self.parameters()).device,)
        if getattr(self, "_pack_sig", None) != sig or getattr(self, "_packs", None) is None:
            packs = {}
            for n, p in self.named_parameters():
                if p.dim() != 2 or n.startswith("embedding"):
                    continue
                pk = WeightPack()
                pk.n, pk.k = p.shape
                old = getattr(self, "_packs", None)
                if old is not None and n in old and old[n].w16.shape == p.shape:      # keep the buffers (addresses stay valid)
                    pk.w16, pk.w16_t = old[n].w16, old[n].w16_t
                    pk.w16.copy_(p.detach())
                else:
                    pk.w16 = p.detach().to(BF16).contiguous()
                    pk.w16_t = torch.empty((pk.k, pk.n), dtype=BF16, device=p.device)
                check(lib.mi355_transpose_bf16(pk.w16.data_ptr(), pk.w16_t.data_ptr(), 1, pk.n, pk.k, _stream()), "transpose")
                packs[n] = pk
            self._packs, self._pack_sig = packs, sig
        return self._packs

    def operands_current(self):
        """Tell the module its bf16 operand packs already match the fp32 masters (a fused optimizer wrote both)."""
        self._pack_sig = tuple(p._version for p in self.parameters()) + (next(self.parameters()).device,)

    def _forward_train_hip(self, input_ids, attention_mask, return_pooled):
        """Every row computed, dropout active when `self.training`: the reference's training arithmetic
        (encoder/text_encoder.py:29-43 under train_encoder.py:103-105) on the library's kernels, differentiable."""
        from .ops import LinearFn, LayerNormFn
        packs = self._train_packs()
        B, T = input_ids.shape
        D = self.embedding.embedding_dim
        H = self.transformer.layers[0].self_attn.num_heads
        p = float(self.transformer.layers[0].p) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,))) if p > 0 else 0     # CPU generator: no device synchronisation
        kv_len, bad = self._key_counts(attention_mask)
        x = _ToBf16Fn.apply(self.embedding(input_ids)).view(B * T, D)
        for i, layer in enumerate(self.transformer.layers):
            pre = f"transformer.layers.{i}."
            sa = layer.self_attn
            s0 = (seed + 7919 * (4 * i + 1)) & 0x7FFFFFFF
            qkv = LinearFn.apply(x, packs[pre + "self_attn.in_proj_weight"], sa.in_proj_bias, None, False, sa.in_proj_weight)
            a = _AttnTrainFn.apply(qkv.view(B, T, 3 * D), kv_len, H, p, s0).view(B * T, D)
            if p > 0:
                a = LinearFn.apply(a, packs[pre + "self_attn.out_proj.weight"], sa.out_proj.bias, None, False, sa.out_proj.weight)
                h = _DropoutAddFn.apply(a, x, p, s0 + 1)
            else:
                h = LinearFn.apply(a, packs[pre + "self_attn.out_proj.weight"], sa.out_proj.bias, x, False, sa.out_proj.weight)
            x = LayerNormFn.apply(h, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            f = LinearFn.apply(x, packs[pre + "linear1.weight"], layer.linear1.bias, None, False, layer.linear1.weight)
            f = _GeluDropoutFn.apply(f, p, s0 + 2)
            if p > 0:
                f = LinearFn.apply(f, packs[pre + "linear2.weight"], layer.linear2.bias, None, False, layer.linear2.weight)
                h = _DropoutAddFn.apply(f, x, p, s0 + 3)
            else:
                h = LinearFn.apply(f, packs[pre + "linear2.weight"], layer.linear2.bias, x, False, layer.linear2.weight)
            x = LayerNormFn.apply(h, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        if return_pooled:       # LayerNorm and projection are per row: only the first token's rows are needed
            x0 = x.view(B, T, D)[:, 0].contiguous()
            x0 = LayerNormFn.apply(x0, self.norm.weight, self.norm.bias, self.norm.eps)
            y = LinearFn.apply(x0, packs["projection.weight"], self.projection.bias, None, True, self.projection.weight)
            return self._poison(_L2NormFn.apply(y), bad)
        x = LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        y = LinearFn.apply(x, packs["projection.weight"], self.projection.bias, None, True, self.projection.weight)
        return self._poison(y.view(B, T, -1), bad)

    def _require_kernels(self, input_ids):
        d = self.embedding.embedding_dim
        h = self.transformer.layers[0].self_attn.num_heads
        if not input_ids.is_cuda:
            raise RuntimeError("TextEncoder: inputs and parameters must live on the HIP device (there is no CPU path)")
        if d % 8 or d % h or (d // h) % 8 or d // h > 96:
            raise ValueError(f"TextEncoder: embed_dim {d} with {h} heads is not carried by the kernels (the width and the "
                             "head width must be multiples of 8, heads at most 96 wide); there is no other backend")

    def forward(self, input_ids, attention_mask, return_pooled=True):
        self._require_kernels(input_ids)
        train_path = torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters()))
        if train_path or return_pooled:
            return self._forward_train_hip(input_ids, attention_mask, return_pooled)
        return self._forward_hip(input_ids, attention_mask)
