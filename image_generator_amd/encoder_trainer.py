"""Contrastive pre-training step of the text encoder: the build's counterpart of the hot loop in
/root/reference/encoder/train_encoder.py:83-110.

Step order kept from the reference: tokenised captions -> `TextEncoder` in training mode (pooled, L2-normalised) ->
`ContrastiveLoss` against the image embeddings -> backward -> AdamW (train_encoder.py:42-49: lr for the encoder, 0.1 lr
for the loss's log-temperature, torch's default weight decay 1e-2, no gradient clipping).

What is different on purpose:
  * the image tower (`encoder/image_encoder.py`: a pretrained CLIP ViT fetched by name) is not part of this package; the
    step takes its output, the image embeddings, as an input and returns their gradient for whoever owns that tower;
  * bf16 compute with fp32 masters instead of fp16 autocast + GradScaler (train_encoder.py:51,103-109);
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
        self.state = torch.zeros(2, dtype=F32, device=dev)       # [0] stays 0: no clipping, the norm is not needed
        self._hyper_host = None
        # the loss's single scalar: torch's own AdamW (one element; the library's update works on multiples of 4)
        self.temp_opt = torch.optim.AdamW(criterion.parameters(), lr=lr * 0.1, betas=betas, eps=eps, weight_decay=weight_decay)
        self.step_count = 0

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
        self.temp_opt.zero_grad(set_to_none=True)
        text = self.enc(input_ids, attention_mask, return_pooled=True)
        loss = self.crit(text, img)
        loss.backward()
        self._set_hyper()
        packs = self.enc._train_packs()
        names = {id(p): n for n, p in self.enc.named_parameters()}
        s = _stream()
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad if (p.grad.dtype == F32 and p.grad.is_contiguous()) else p.grad.to(F32).contiguous()
            pk = packs.get(names[id(p)])
            p16 = None if pk is None else pk.w16.data_ptr()
            check(lib.mi355_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p16, self.hyper.data_ptr(),
                                       self.state.data_ptr(), p.numel(), s), "adamw")
        for pk in packs.values():        # the data-gradient operands of the next step
            check(lib.mi355_transpose_bf16(pk.w16.data_ptr(), pk.w16_t.data_ptr(), 1, pk.n, pk.k, s), "transpose")
        self.enc.operands_current()
        self.temp_opt.step()
        self.step_count += 1
        return loss.detach(), img.grad
