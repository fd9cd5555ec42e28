"""Low-rank adapters with the interface of /root/reference/lora.py:5-41.

`LoRALinear(original_linear, r=4, alpha=1.0)` wraps either a torch `nn.Linear` living on the HIP
device or one of this package's `HipLinear` layers.  y = W x + b + alpha * B (A x) is ONE MFMA
GEMM: the rank-r product enters through the GEMM epilogue (`t = x A^T` is a skinny side product),
the backward produces dx, dA, dB only (the frozen base weight gets no gradient).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._lib import lib, check
from . import ops as _ops
from .ops import BF16, F32, _ptr, _stream, gemm, gemm_skinny


def _operands(original):
    """(w16 [N][K], w16_t [K][N]) bf16 copies of the base weight, cached until the weight changes."""
    pack = getattr(original, "pack", None)
    if pack is not None and pack.w16 is not None and pack.w16_t is not None:
        return pack.w16, pack.w16_t
    w = original.weight
    key = (w.data_ptr(), w._version, w.device)
    cache = getattr(original, "_mi355_lora_cache", None)
    if cache is None or cache[0] != key:
        w16 = w.detach().to(BF16).contiguous()
        w16_t = torch.empty((w.shape[1], w.shape[0]), dtype=BF16, device=w.device)
        check(lib.mi355_transpose_bf16(w16.data_ptr(), w16_t.data_ptr(), 1, w.shape[0], w.shape[1], _stream()), "transpose")
        cache = (key, w16, w16_t)
        original._mi355_lora_cache = cache
    return cache[1], cache[2]


def _f32c(p):
    p = p.detach()
    return p if (p.dtype == F32 and p.is_contiguous()) else p.to(F32).contiguous()


class _LoraLinearFn(torch.autograd.Function):
    """y = x W^T + b + alpha * (x A^T) B^T (+ residual).

    `gdst` = (dA_pad [8][K], dB_pad [N][8], A16 [8][K], Bt16 [8][N]): the direct-gradient mode of a DenoiserTrainer
    step.  A16 / Bt16 are per-step bf16 copies of the factors, zero-padded to 8 ranks: t = alpha x A^T and
    dt = alpha dy B are skinny MFMA GEMMs on them (bf16 [M][8] results that the base GEMMs' epilogues read), and the
    adapter gradients dB = dy^T t, dA = dt^T x are two more jobs of the grouped MFMA weight-gradient launch, written
    rank-padded into the fp32 scratch views; the owner compacts them into the flat gradient buffer once per step.
    Without `gdst` the fp32 side kernels run and the gradients are returned to autograd."""

    @staticmethod
    def forward(ctx, x, a, b, bias, alpha, w16, w16_t, residual, gdst):
        if not x.is_cuda:
            raise RuntimeError("LoRALinear: HIP device tensors required (no CPU path)")
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.dtype != BF16 or not x2.is_contiguous():
            x2 = x2.to(BF16).contiguous()
        M, r, N = x2.shape[0], a.shape[0], b.shape[0]
        af, bf = _f32c(a), _f32c(b)
        biasf = None if bias is None else _f32c(bias)
        res2 = None if residual is None else residual.reshape(-1, N)
        queued = gdst is not None and r <= 8 and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if queued:
            # training step: t = alpha * x A^T as a skinny MFMA GEMM on the per-step bf16 copy of A (zero-padded to 8
            # ranks); the same bf16 matrix feeds the epilogue below and the weight-gradient job for dB
            t = None
            t16 = gemm_skinny(x2, gdst[2], alpha)
            y = gemm(x2, w16, bias=biasf, residual=res2, lora_t=t16, lora_b=bf, lora_alpha=1.0)
        else:
            t16 = None
            t = torch.empty((M, r), dtype=F32, device=x.device)
            check(lib.mi355_lora_down(x2.data_ptr(), af.data_ptr(), t.data_ptr(), M, K, r, None, alpha, _stream()), "lora_down")
            y = gemm(x2, w16, bias=biasf, residual=res2, lora_t=t, lora_b=bf, lora_alpha=alpha)
        ctx.save_for_backward(x2, t, af, bf, t16)
        ctx.alpha, ctx.w16_t, ctx.xshape, ctx.xdtype, ctx.gdst = alpha, w16_t, x.shape, x.dtype, gdst
        y = y.view(*x.shape[:-1], N)
        return y if x.dtype == BF16 else y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x2, t, af, bf, t16 = ctx.saved_tensors
        M, K = x2.shape
        N, r = bf.shape
        dy2 = dy.reshape(-1, N)
        if dy2.dtype != BF16 or not dy2.is_contiguous():
            dy2 = dy2.to(BF16).contiguous()
        s = _stream()
        fast = t16 is not None
        if fast and _ops._WGRAD_QUEUE is None:
            raise RuntimeError("LoRALinear: direct-gradient mode needs the trainer's weight-gradient queue")
        queue = _ops._WGRAD_QUEUE if fast else None
        if fast:
            dt = dt16 = gemm_skinny(dy2, ctx.gdst[3], ctx.alpha)          # alpha * dy B as a skinny GEMM, bf16 [M][8]
        else:
            dt = torch.empty((M, r), dtype=F32, device=dy.device)
            check(lib.mi355_lora_dt(dy2.data_ptr(), bf.data_ptr(), dt.data_ptr(), M, N, r, 1, ctx.alpha, None, s), "lora_dt")
        da = db = None
        want_a, want_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if queue is not None:
            ga, gb = ctx.gdst[0], ctx.gdst[1]
            if want_b:          # dB_pad[n][0:8] = sum_m dy[m][n] * (alpha t)[m][0:8]
                queue.push(dy2, t16, gb, None, M, 1, 1, 8, N, 1, 1, False, 1, 1)
            if want_a:          # dA_pad[0:8][k] = sum_m dt[m][0:8] * x[m][k]
                queue.push(dt16, x2, ga, None, M, 1, 1, K, 8, 1, 1, False, 1, 1)
        else:
            if want_b:
                db = torch.empty((N, r), dtype=F32, device=dy.device)
                check(lib.mi355_lora_outer(dy2.data_ptr(), t.data_ptr(), db.data_ptr(), M, N, r, ctx.alpha, 0, s), "lora_outer")
            if want_a:
                da = torch.empty((r, K), dtype=F32, device=dy.device)
                check(lib.mi355_lora_outer(x2.data_ptr(), dt.data_ptr(), da.data_ptr(), M, K, r, 1.0, 1, s), "lora_outer")
        dx = None
        if ctx.needs_input_grad[0]:
            # dx = dy W + dt A : the rank-r term rides in the epilogue of the data-gradient GEMM (A as stored, [r][K])
            dx = gemm(dy2, ctx.w16_t, lora_t=dt, lora_b=af, lora_alpha=1.0, lora_rmajor=True)
            dx = dx.view(ctx.xshape)
            if ctx.xdtype != BF16:
                dx = dx.to(ctx.xdtype)
        dres = dy if ctx.needs_input_grad[7] else None
        return dx, da, db, None, None, None, None, dres, None


class LoraFusedFn(torch.autograd.Function):
    """Row-stacked projections that read the same input (q/k/v of self-attention, k/v of cross-attention), each with
    its own rank-4 adapter, as ONE GEMM: y[:, s] = x W_s^T + alpha * (x A_s^T) B_s^T.

    `lf` carries zero-copy stacked views of the adapters' factors (`A_all` [S*r][K], `B_all` [S*C][r]; the owning
    model lays the parameters out back to back) and, in direct-gradient mode,
    `grad_dst` = (dA_pad [W][K], dB_pad [S*C][W], A16 [W][K], Bt16 [W][S*C]), W = 8 or 16: the rank-padded gradient
    scratch views and the per-step bf16 operands of the skinny GEMMs (Bt16 block-diagonal).  `params` = the 2*S factor parameters
    (A_0, B_0, A_1, B_1, ...) for autograd's bookkeeping."""

    @classmethod
    def apply_to(cls, x, pack, lf):
        """Call site helper: the per-step bf16 operands of direct-gradient mode are used only inside a trainer step
        that records gradients (see LoRALinear.forward)."""
        return cls.apply(x, pack, lf, torch.is_grad_enabled() and _ops._DIRECT_STEP, *lf.params)

    @staticmethod
    def forward(ctx, x, pack, lf, direct, *params):
        if not x.is_cuda:
            raise RuntimeError("LoRA: HIP device tensors required (no CPU path)")
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.dtype != BF16 or not x2.is_contiguous():
            x2 = x2.to(BF16).contiguous()
        M, N, RT = x2.shape[0], pack.n, lf.S * lf.r
        want = any(ctx.needs_input_grad[4:])
        W = 8 if RT <= 8 else 16
        fast = want and direct and lf.grad_dst is not None
        if fast:        # skinny MFMA GEMM on the per-step bf16 copy of the stacked A factors (see _LoraLinearFn)
            t16 = gemm_skinny(x2, lf.grad_dst[2], lf.alpha)
            y = gemm(x2, pack.w16.view(N, K), lora_t=t16, lora_b=lf.B_all, lora_alpha=1.0, lora_seg=N // lf.S)
        else:
            t = torch.empty((M, RT), dtype=F32, device=x.device)
            t16 = torch.empty((M, W), dtype=BF16, device=x.device) if want else None
            check(lib.mi355_lora_down(x2.data_ptr(), lf.A_all.data_ptr(), t.data_ptr(), M, K, RT, _ptr(t16), lf.alpha, _stream()),
                  "lora_down")
            y = gemm(x2, pack.w16.view(N, K), lora_t=t, lora_b=lf.B_all, lora_alpha=lf.alpha, lora_seg=N // lf.S)
        ctx.save_for_backward(x2, t16)
        ctx.pack, ctx.lf, ctx.xshape, ctx.W, ctx.fast = pack, lf, x.shape, W, fast
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, t16 = ctx.saved_tensors
        pack, lf, W = ctx.pack, ctx.lf, ctx.W
        M, K = x2.shape
        N, S, r = pack.n, lf.S, lf.r
        C = N // S
        dy2 = dy.reshape(-1, N)
        if dy2.dtype != BF16 or not dy2.is_contiguous():
            dy2 = dy2.to(BF16).contiguous()
        want = t16 is not None
        if ctx.fast:
            if _ops._WGRAD_QUEUE is None:
                raise RuntimeError("LoRA: direct-gradient mode needs the trainer's weight-gradient queue")
            dt = dt16 = gemm_skinny(dy2, lf.grad_dst[3], lf.alpha)       # block-diagonal B^T operand: dt[:, 4s+q] = alpha dy_s B_s
        else:
            dt = torch.empty((M, S * r), dtype=F32, device=dy.device)
            dt16 = torch.empty((M, W), dtype=BF16, device=dy.device) if want else None
            check(lib.mi355_lora_dt(dy2.data_ptr(), lf.B_all.data_ptr(), dt.data_ptr(), M, N, r, S, lf.alpha, _ptr(dt16), _stream()),
                  "lora_dt")
        grads = [None] * (2 * S)
        if want:
            queue = _ops._WGRAD_QUEUE if ctx.fast else None
            if queue is not None:
                ga, gb = lf.grad_dst[0], lf.grad_dst[1]
                queue.push(dy2, t16, gb, None, M, 1, 1, W, N, 1, 1, False, 1, 1)       # dB_pad = dy^T (alpha t)
                queue.push(dt16, x2, ga, None, M, 1, 1, K, W, 1, 1, False, 1, 1)       # dA_pad = dt^T x
            else:
                gb = _ops.wgrad(dy2, t16, W, N).view(N, W)
                ga = _ops.wgrad(dt16, x2, K, W).view(W, K)
                for i in range(S):
                    if ctx.needs_input_grad[4 + 2 * i]:
                        grads[2 * i] = ga[i * r:(i + 1) * r]
                    if ctx.needs_input_grad[5 + 2 * i]:
                        grads[2 * i + 1] = gb[i * C:(i + 1) * C, i * r:(i + 1) * r]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy2, pack.w16_t.view(K, N), lora_t=dt, lora_b=lf.A_all, lora_alpha=1.0, lora_rmajor=True).view(ctx.xshape)
        return (dx, None, None, None, *grads)


class LoRALinear(nn.Module):
    def __init__(self, original_linear, r=4, alpha=1.0):
        super().__init__()
        self.original = original_linear
        self.r = r
        self.alpha = alpha
        dev = original_linear.weight.device
        self.lora_A = nn.Linear(original_linear.in_features, r, bias=False, device=dev)
        self.lora_B = nn.Linear(r, original_linear.out_features, bias=False, device=dev)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.weight)
        for p in self.original.parameters():
            p.requires_grad = False
        self.grad_dst = None        # (dA_pad [8][K], dB_pad [N][8]) scratch views in direct-gradient mode

    def forward(self, x, residual=None):
        """`residual` (optional, beyond the reference signature) is added in the same GEMM epilogue."""
        w16, w16_t = _operands(self.original)
        # The direct-gradient operands (bf16 copies of the factors, packed at the start of each trainer step) belong to
        # that step only: any other forward pass (evaluation or sampling between steps, plain autograd) reads the fp32
        # factors themselves.  Decided here, because grad mode is always off inside Function.forward and
        # `ctx.needs_input_grad` does not depend on it.
        gdst = self.grad_dst if (torch.is_grad_enabled() and _ops._DIRECT_STEP) else None
        return _LoraLinearFn.apply(x, self.lora_A.weight, self.lora_B.weight, self.original.bias, self.alpha, w16, w16_t,
                                   residual, gdst)


def extract_lora_weights(model):
    out = {}
    for name, mod in model.named_modules():
        if isinstance(mod, LoRALinear):
            out[f"{name}.lora_A"] = mod.lora_A.state_dict()
            out[f"{name}.lora_B"] = mod.lora_B.state_dict()
    return out


def load_lora_weights(model, lora_weights):
    for name, mod in model.named_modules():
        if isinstance(mod, LoRALinear) and f"{name}.lora_A" in lora_weights:
            mod.lora_A.load_state_dict(lora_weights[f"{name}.lora_A"])
            mod.lora_B.load_state_dict(lora_weights[f"{name}.lora_B"])
