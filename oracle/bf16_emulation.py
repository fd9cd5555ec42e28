"""Test infrastructure (never imported by the product): the fp32 CPU oracle re-run with the HIP path's storage
precision - GEMM weights and every activation that the kernels write to HBM (outputs of convolutions, linear layers,
GroupNorm and LayerNorm) rounded to bf16, gradients of those activations rounded to bf16 on the way back, all arithmetic
still fp32.  The distance between this run's gradients and the plain fp32 oracle's is the noise floor that bf16 storage
alone produces; the GPU parity tests set their thresholds from it instead of from a guess."""
import copy

import torch
import torch.nn as nn


class RoundBF16(torch.autograd.Function):
    """y = bf16(x) in the forward pass, dx = bf16(dy) in the backward pass (straight-through rounding)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def emulate_bf16_storage(model):
    """A deep copy of `model` (an fp32 torch module) with bf16-rounded weight matrices and rounding hooks behind every
    Conv2d / Linear / GroupNorm / LayerNorm."""
    emu = copy.deepcopy(model)
    with torch.no_grad():
        for p in emu.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    for m in emu.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear, nn.GroupNorm, nn.LayerNorm)):
            m.register_forward_hook(lambda mod, args, out: RoundBF16.apply(out))
    return emu


def gradient_distances(model, reference):
    """{name: (cosine, relative L2)} of `model`'s parameter gradients against `reference`'s (parameters whose reference
    gradient vanishes are left out)."""
    ref = dict(reference.named_parameters())
    out = {}
    for n, p in model.named_parameters():
        a, b = p.grad.detach().float().cpu().flatten().double(), ref[n].grad.detach().float().cpu().flatten().double()
        if float(b.norm()) < 1e-12:
            continue
        out[n] = (float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)), float((a - b).norm() / b.norm()))
    return out
