"""Training objectives on the HIP device.

`ContrastiveLoss` keeps the interface of /root/reference/loss.py:6-28 (ctor argument, `log_temp`
parameter, `temperature` property, forward(text_embeds, image_embeds)); the arithmetic - L2
normalisation, similarity logits, both cross-entropies and every gradient - runs in one fused
library call.  `mse_loss` is the denoiser objective of generator/image_generator.py:125.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._lib import lib, check
from .ops import F32, _stream, mse_loss  # noqa: F401  (re-exported)


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, image, log_temp):
        if not text.is_cuda:
            raise RuntimeError("ContrastiveLoss: HIP device tensors required (no CPU path)")
        t = text.to(F32).contiguous()
        i = image.to(F32).contiguous()
        lt = log_temp.detach().to(F32).reshape(1).contiguous()
        B, D = t.shape
        loss = torch.empty((), dtype=F32, device=t.device)
        dt, di = torch.empty_like(t), torch.empty_like(i)
        dlt = torch.empty(1, dtype=F32, device=t.device)
        scratch = torch.empty(int(lib.mi355_contrastive_scratch_floats(B, D)), dtype=F32, device=t.device)
        check(lib.mi355_contrastive_loss_fused(t.data_ptr(), i.data_ptr(), lt.data_ptr(), loss.data_ptr(), dt.data_ptr(),
                                               di.data_ptr(), dlt.data_ptr(), scratch.data_ptr(), B, D, _stream()),
              "contrastive_loss")
        ctx.save_for_backward(dt, di, dlt)
        ctx.dtypes = (text.dtype, image.dtype, log_temp.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        dt, di, dlt = ctx.saved_tensors
        return (dt * g).to(ctx.dtypes[0]), (di * g).to(ctx.dtypes[1]), (dlt * g).reshape(ctx.dtypes[2])


class ContrastiveLoss(nn.Module):
    def __init__(self, init_temp=0.07):
        super().__init__()
        self.log_temp = nn.Parameter(torch.tensor(math.log(init_temp), dtype=torch.float32))

    @property
    def temperature(self) -> torch.Tensor:
        return self.log_temp.exp().clamp(1e-4, 1e4)

    def forward(self, text_embeds, image_embeds):
        return _ContrastiveFn.apply(text_embeds, image_embeds, self.log_temp)
