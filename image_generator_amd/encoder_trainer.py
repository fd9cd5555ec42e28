"""Contrastive pre-training step of the text encoder: the build's counterpart of the hot loop in
/root/reference/encoder/train_encoder.py:83-110.

Step order kept from the reference: tokenised captions -> `TextEncoder` in training mode (pooled, L2-normalised) ->
`ContrastiveLoss` against the image embeddings -> backward -> AdamW (train_encoder.py:42-49: lr for the encoder, 0.1 lr
for the loss's log-temperature, torch's default weight decay 1e-2, no gradient clipping).

What is different on purpose:
  * the image tower (`encoder/image_encoder.py`: a pretrained CLIP ViT fetched by name) is not part of this package; the
    step takes its output, the image embeddings, as an input and returns their gradient for whoever owns that tower;
  * bf16 compute with fp32 masters instead of fp16 autocast + GradScaler (train_encoder.py:51,103-109).  What the
    GradScaler also does - leave every parameter untouched when a gradient is not finite - is kept: one table-driven
    launch sums the squares of all gradients (the temperature's included), the update kernels skip on a non-finite sum
    and count the skip, and the temperature's own update is gated by the same device flag.  No host synchronisation;
  * every matrix product, the attention (mask and dropout inside the flash kernels), the normalisations, the loss and
    the parameter update run in libmi355_denoiser.so; the update is `mi355_adamw_step` per parameter tensor, which also
    writes the bf16 operand copy the next forward pass multiplies with.
"""
from __future__ import annotations

import torch

from ._lib import lib, check
from .ops import F32, _stream


class ContrastiveTrainer:
    def __init__(self, text_encoder, criterion, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.enc, self.crit = text_encoder, criterion
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        dev = next(text_encoder.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("ContrastiveTrainer: parameters must live on the HIP device (no CPU path)")
        self.params = [p for p in text_encoder.parameters() if p.requires_grad]
        if any(p.numel() % 4 for p in self.params):
            raise ValueError("ContrastiveTrainer: parameter sizes must be multiples of 4")
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.hyper = torch.zeros(9, dtype=F32, device=dev)
        # [0] = sum of squared gradients of the step (no clipping: only its finiteness is used), [1] = skip count
        self.state = torch.zeros(4, dtype=F32, device=dev)
        self._hyper_host = None
        self._table_host = None
        self._table = torch.zeros((len(self.params) + 1, 2), dtype=torch.int64, device=dev)
        # the loss's single scalar (the library's update works on multiples of 4): AdamW in a few one-element tensor
        # ops, lr 0.1 x the encoder's (train_encoder.py:42-49), gated by the same non-finite flag
        self.temp = [p for p in criterion.parameters() if p.requires_grad]
        self.temp_m = [torch.zeros_like(p) for p in self.temp]
        self.temp_v = [torch.zeros_like(p) for p in self.temp]
        self.step_count = 0
        self._updates_per_step = 1

    def _set_hyper(self):
        from .ops import PinnedRing
        if self._hyper_host is None:
            self._hyper_host = PinnedRing(lambda: torch.zeros(9, dtype=F32).pin_memory(), 8)
        self.t += 1
        h = self._hyper_host.acquire()
        h[0], h[1], h[2], h[3], h[4] = self.lr, self.betas[0], self.betas[1], self.eps, self.wd
        h[5], h[6] = 1 - self.betas[0] ** self.t, 1 - self.betas[1] ** self.t
        h[7], h[8] = 0.0, 1.0
        self.hyper.copy_(h, non_blocking=True)
        self._hyper_host.copied()
        self._hyper_host.release()

    def train_step(self, input_ids, attention_mask, image_embeds):
        """Returns (loss, gradient of the loss with respect to image_embeds)."""
        self.enc.train()
        img = image_embeds.detach().to(F32).requires_grad_(True)
        for p in self.params:
            p.grad = None
        for p in self.temp:
            p.grad = None
        text = self.enc(input_ids, attention_mask, return_pooled=True)
        loss = self.crit(text, img)
        loss.backward()
        self._set_hyper()
        packs = self.enc._train_packs()
        names = {id(p): n for n, p in self.enc.named_parameters()}
        s = _stream()
        grads = []
        for p in self.params:
            g = p.grad
            if g is not None and not (g.dtype == F32 and g.is_contiguous()):
                g = g.to(F32).contiguous()
            grads.append(g)
        tgrads = [None if p.grad is None else p.grad.to(F32).contiguous() for p in self.temp]
        self._sum_squares([g for g in grads + tgrads if g is not None], s)
        n_upd = 0
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if g is None:
                continue
            pk = packs.get(names[id(p)])
            p16 = None if pk is None else pk.w16.data_ptr()
            check(lib.mi355_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p16, self.hyper.data_ptr(),
                                       self.state.data_ptr(), p.numel(), s), "adamw")
            n_upd += 1
        self._updates_per_step = max(n_upd, 1)
        for pk in packs.values():        # the data-gradient operands of the next step
            check(lib.mi355_transpose_bf16(pk.w16.data_ptr(), pk.w16_t.data_ptr(), 1, pk.n, pk.k, s), "transpose")
        self.enc.operands_current()
        self._temperature_step(tgrads)
        self.step_count += 1
        return loss.detach(), img.grad

    def _sum_squares(self, grads, s):
        """state[0] = sum of the squares of every gradient of this step: one table (address, count) through a pinned
        ring, one launch."""
        from .ops import PinnedRing
        n = len(grads)
        if self._table_host is None:
            rows = self._table.shape[0]
            self._table_host = PinnedRing(lambda: torch.zeros((rows, 2), dtype=torch.int64).pin_memory(), 8)
        h = self._table_host.acquire()
        for i, g in enumerate(grads):
            h[i, 0], h[i, 1] = g.data_ptr(), g.numel()
        self._table.copy_(h, non_blocking=True)
        self._table_host.copied()
        self._table_host.release()
        check(lib.mi355_zero_f32(self.state.data_ptr(), 1, s), "zero")
        if n:
            check(lib.mi355_sumsq_ranges(self._table.data_ptr(), n, self.state.data_ptr(), s), "sumsq_ranges")

    def _temperature_step(self, tgrads):
        b1, b2 = self.betas
        lr = self.lr * 0.1
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        ok = torch.isfinite(self.state[0])                      # device flag: nothing moves on a non-finite step
        with torch.no_grad():
            for p, g, m, v in zip(self.temp, tgrads, self.temp_m, self.temp_v):
                if g is None:
                    continue
                g = torch.where(ok, g, torch.zeros_like(g))
                m_new = b1 * m + (1 - b1) * g
                v_new = b2 * v + (1 - b2) * g * g
                p_new = p * (1 - lr * self.wd) - lr * (m_new / bc1) / ((v_new / bc2).sqrt() + self.eps)
                m.copy_(torch.where(ok, m_new, m))
                v.copy_(torch.where(ok, v_new, v))
                p.copy_(torch.where(ok, p_new, p))

    def skipped_steps(self) -> int:
        """Steps whose gradients were not finite and that therefore changed nothing (host synchronisation)."""
        return int(round(float(self.state[1]) / self._updates_per_step))
