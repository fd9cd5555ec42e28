"""`ImageGenerator`: drop-in for /root/reference/generator/image_generator.py:13-126 on MI355X.

Same constructor call (`ImageGenerator(device)`), attributes (`device`, `dtype`, `tokenizer`,
`text_encoder`, `unet`, `vae`, `scheduler`, `infer_scheduler`) and methods (`encode_text`,
`decode_latents`, `forward`, `train_step`) as the reference, so trainer.py / train_generator.py /
inference.py keep working:  `model.unet = wrap(model.unet)`, `model.train_step(images, text)`,
`model(sample_text)`.

What differs, and why:
  * the reference downloads six pretrained artefacts by name (image_generator.py:19-37); nothing can
    be fetched here, so the parts are constructed from configuration objects with random
    initialisation and can be filled from checkpoints with `load_state_dict` (parameter names are
    checkpoint-compatible for the denoiser and for the reference's own TextEncoder);
  * conditioning comes from the repository's `TextEncoder` in its sequence mode
    (encoder/text_encoder.py:41-43, 768-wide = the denoiser's cross-attention width);
  * `vae` is the SD-v1 `AutoencoderKL` architecture on the same HIP kernels (vae.py), frozen and
    inference-only as in the reference (image_generator.py:40), random-initialised until real weights
    are loaded.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .schedulers import DDIMScheduler, DDPMScheduler
from .text_encoder import TextEncoder
from .tokenizer import HashTokenizer, load_tokenizer
from .unet import DenoiserConfig, UNet2DConditionModel
from .vae import AutoencoderKL, VaeConfig


class ImageGenerator(nn.Module):
    def __init__(self, device, tokenizer_path=None, unet_config: DenoiserConfig | None = None,
                 text_encoder_kwargs: dict | None = None, vae_config: VaeConfig | None = None):
        super().__init__()
        self.device = device
        self.dtype = torch.float32
        te_kwargs = dict(text_encoder_kwargs or {})
        self.tokenizer = load_tokenizer(tokenizer_path) if tokenizer_path else HashTokenizer(te_kwargs.get("vocab_size", 16384))
        cfg = unet_config or DenoiserConfig.sd_v1()
        te_kwargs.setdefault("projection_dim", cfg.cross_attention_dim)
        self.text_encoder = TextEncoder(**te_kwargs).to(self.device)
        self.unet = UNet2DConditionModel(cfg).to(self.device).to(self.dtype)
        if vae_config is None:        # the SD-v1 auto-encoder next to the SD-v1 denoiser, a small one next to a test-size denoiser
            vae_config = VaeConfig.sd_v1() if unet_config is None else VaeConfig.small()
        self.vae = AutoencoderKL(vae_config).to(self.device).to(self.dtype)
        self.scheduler = DDPMScheduler()
        self.infer_scheduler = DDIMScheduler()
        self.text_encoder.requires_grad_(False)
        self.text_encoder.eval()
        self.vae.requires_grad_(False)
        self._step_graph = None

    def train(self, mode=True):
        super().train(mode)
        self.text_encoder.eval()      # frozen conditioning encoder: always the inference path
        return self

    # image_generator.py:42-61
    def encode_text(self, text, max_length=77, padding="max_length"):
        encoder_device = next(self.text_encoder.parameters()).device
        if isinstance(text, str):
            text = [text]
        text_inputs = self.tokenizer(text, padding=padding, max_length=max_length, truncation=True,
                                     return_tensors="pt").to(encoder_device)
        with torch.no_grad():
            return self.text_encoder(text_inputs.input_ids, text_inputs.attention_mask, return_pooled=False)

    # image_generator.py:63-71
    def decode_latents(self, latents):
        vae_device = next(self.vae.parameters()).device
        latents = 1 / 0.18215 * latents.to(vae_device)
        with torch.no_grad():
            image = self.vae.decode(latents).sample
        return (image / 2 + 0.5).clamp(0, 1)

    # image_generator.py:73-101 (DDIM, eta 0, classifier-free guidance).  One prompt per call in the
    # reference; a list of prompts here yields one image per prompt with the same per-step arithmetic.
    @torch.no_grad()
    def forward(self, text, num_inference_steps=50, guidance_scale=7.5, latents=None, height=512, width=512,
                use_graph=True):
        unet_device = next(self.unet.parameters()).device
        if isinstance(text, str):
            text = [text]
        n = len(text)
        cond = self.encode_text(text)
        uncond = self.encode_text([""] * n)
        emb = torch.cat([uncond, cond]).to(unet_device).to(torch.bfloat16).contiguous()
        if latents is None:
            latents = torch.randn((n, self.unet.config.in_channels, height // 8, width // 8), device=unet_device,
                                  dtype=self.dtype)
        latents = self.sample_latents(emb, latents, num_inference_steps, guidance_scale, use_graph)
        return self.decode_latents(latents)

    @torch.no_grad()
    def sample_latents(self, emb_pair, latents, num_inference_steps=50, guidance_scale=7.5, use_graph=True):
        """The 50-iteration hot loop.  Per step: duplicate latents -> denoiser -> fused guidance + DDIM
        update.  With `use_graph` one step is captured into a HIP graph (static shapes; the timestep and
        the schedule coefficients are device scalars refreshed before every replay)."""
        sch = self.infer_scheduler
        sch.set_timesteps(num_inference_steps)
        dev = latents.device
        n = latents.shape[0]
        lat = latents.to(self.dtype).contiguous().clone()
        t_dev = torch.zeros(2 * n, dtype=torch.float32, device=dev)
        coef_dev = torch.zeros(3, dtype=torch.float32, device=dev)
        host = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(len(sch.timesteps))]

        def one_step():
            x2 = torch.cat([lat, lat])
            eps = self.unet(x2, t_dev, encoder_hidden_states=emb_pair).sample
            lat.copy_(sch.step_cfg(eps, lat, coef_dev))

        from .unet import context_kv_cache
        with context_kv_cache():            # keys / values of the text states: projected in the first step only
            return self._sampling_loop(sch, one_step, lat, emb_pair, t_dev, coef_dev, host, guidance_scale, n, use_graph)

    def _sampling_loop(self, sch, one_step, lat, emb_pair, t_dev, coef_dev, host, guidance_scale, n, use_graph):
        graph = None
        for i, t in enumerate(sch.timesteps):
            a_t, a_p, g = sch.coefficients(t, guidance_scale)
            h = host[i]
            h[0], h[1], h[2], h[3] = a_t, a_p, g, float(t)
            coef_dev.copy_(h[:3], non_blocking=True)
            t_dev.copy_(h[3:4].expand(2 * n), non_blocking=True)
            if not use_graph:
                one_step()
            elif graph is None:
                one_step()                                   # warm-up (allocations, operand refresh)
                torch.cuda.synchronize()
                key = (tuple(lat.shape), tuple(emb_pair.shape))
                graph = torch.cuda.CUDAGraph()
                # re-run this step inside the capture: restore the latent first
                lat_backup = lat.clone()
                with torch.cuda.graph(graph):
                    one_step()
                lat.copy_(lat_backup)                        # capture does not execute; state is as after warm-up
            else:
                graph.replay()
        return lat

    # image_generator.py:103-126
    def train_step(self, image, text, cond_dropout_prob=0.1):
        unet_device = next(self.unet.parameters()).device
        with torch.no_grad():
            latent = self.vae.encode(image.to(unet_device, self.dtype)).latent_dist.sample()
            latent = latent * 0.18215
        bsz = latent.shape[0]
        t = torch.randint(0, self.scheduler.config.num_train_timesteps, (bsz,), device=unet_device).long()
        noise = torch.randn_like(latent)
        noisy_latent = self.scheduler.add_noise(latent, noise, t)
        if torch.rand(1).item() < cond_dropout_prob:        # whole-batch caption drop, CPU RNG (reference :118)
            text = [""] * bsz
        text_embed = self.encode_text(text, padding="longest").to(unet_device)
        scaled_latent = self.scheduler.scale_model_input(noisy_latent, t)
        noise_pred = self.unet(scaled_latent, t, encoder_hidden_states=text_embed).sample
        return ops.mse_loss(noise_pred, noise)


def inject_lora(unet, r=4, alpha=1.0, targets=("to_q", "to_k", "to_v", "to_out.0")):
    """Freeze the denoiser and wrap its attention projections with LoRALinear, by attribute assignment
    as the reference does for its vision tower (/root/reference/encoder/image_encoder.py:13-19)."""
    from .lora import LoRALinear
    from .unet import Attention
    for p in unet.parameters():
        p.requires_grad = False
    adapters = []
    for mod in unet.modules():
        if isinstance(mod, Attention):
            for name in targets:
                if name == "to_out.0":
                    mod.to_out[0] = LoRALinear(mod.to_out[0], r=r, alpha=alpha)
                    adapters.append(mod.to_out[0])
                else:
                    wrapped = LoRALinear(getattr(mod, name), r=r, alpha=alpha)
                    setattr(mod, name, wrapped)
                    adapters.append(wrapped)
    unet._flat = None          # the parameter set changed: rebuild the flat storage on next use
    return adapters
