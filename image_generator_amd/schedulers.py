"""Noise schedules with the duck-typed surface the reference touches
(/root/reference/generator/image_generator.py:34-37,86-99,114-123): `config.num_train_timesteps`,
`add_noise`, `scale_model_input`, `set_timesteps`, `timesteps`, `step(...).prev_sample`.

Constants follow the scheduler configuration of the SD-v1-4 checkpoint the reference names
(scaled-linear betas 0.00085..0.012, 1000 steps, epsilon prediction, DDIM: leading spacing,
steps_offset 1, set_alpha_to_one False, clip_sample False, eta 0).  The elementwise updates run in
the library (mi355_ddpm_add_noise / mi355_ddim_step_cfg).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from ._lib import lib, check
from .ops import F32, _stream


def sd_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _SchedulerBase:
    def __init__(self, num_train_timesteps=1000):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=0.00085, beta_end=0.012,
                                      beta_schedule="scaled_linear", prediction_type="epsilon", steps_offset=1)
        self.alphas_cumprod = sd_alphas_cumprod(num_train_timesteps)
        self._acp_dev = {}

    def _acp(self, device):
        t = self._acp_dev.get(device)
        if t is None:
            t = self.alphas_cumprod.to(device=device, dtype=F32).contiguous()
            self._acp_dev[device] = t
        return t

    def scale_model_input(self, sample, timestep=None):
        return sample


class DDPMScheduler(_SchedulerBase):
    def add_noise(self, original_samples, noise, timesteps):
        x0 = original_samples.to(F32).contiguous()
        nz = noise.to(F32).contiguous()
        if not x0.is_cuda:
            raise RuntimeError("DDPMScheduler.add_noise: HIP device tensors required (no CPU path)")
        t = timesteps.to(device=x0.device, dtype=torch.int64).contiguous()
        out = torch.empty_like(x0)
        check(lib.mi355_ddpm_add_noise(x0.data_ptr(), nz.data_ptr(), t.data_ptr(), self._acp(x0.device).data_ptr(),
                                       out.data_ptr(), x0.shape[0], x0[0].numel(), _stream()), "ddpm_add_noise")
        return out


class DDIMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000):
        super().__init__(num_train_timesteps)
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def coefficients(self, t, guidance=0.0):
        """(alpha_bar_t, alpha_bar_prev, guidance) for the fused update."""
        t = int(t)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.alphas_cumprod[0])
        return a_t, a_p, float(guidance)

    def step(self, model_output, timestep, sample):
        eps = model_output.to(F32).contiguous()
        x = sample.to(F32).contiguous()
        coef = torch.tensor(self.coefficients(timestep), dtype=F32, device=x.device)
        out = torch.empty_like(x)
        check(lib.mi355_ddim_step_cfg(eps.data_ptr(), x.data_ptr(), out.data_ptr(), coef.data_ptr(), x.numel(), 0,
                                      _stream()), "ddim_step")
        return SimpleNamespace(prev_sample=out)

    def step_cfg(self, eps_pair, sample, coef_dev):
        """Guidance combine + DDIM update in one kernel; eps_pair = [uncond | text] stacked on dim 0,
        coef_dev a device tensor {alpha_bar_t, alpha_bar_prev, guidance}."""
        x = sample
        out = torch.empty_like(x)
        check(lib.mi355_ddim_step_cfg(eps_pair.data_ptr(), x.data_ptr(), out.data_ptr(), coef_dev.data_ptr(), x.numel(), 1,
                                      _stream()), "ddim_step_cfg")
        return out
