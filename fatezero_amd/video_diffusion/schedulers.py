"""DDIM scheduler with the diffusers-0.11.1 interface the reference pipelines use:
`set_timesteps`, `timesteps`, `step(...).prev_sample`, `scale_model_input`, `alphas_cumprod`,
`final_alpha_cumprod`, `config.num_train_timesteps`, `num_inference_steps`, `init_noise_sigma`, `order`.
Defaults are SD-1.x's scheduler_config.json with the two fix-ups the reference forces
(stable_diffusion.py:56-81: steps_offset = 1, clip_sample = False)."""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if prediction_type != "epsilon" or clip_sample:
            raise NotImplementedError("FateZero runs epsilon prediction without sample clipping")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=False, set_alpha_to_one=set_alpha_to_one,
                                      steps_offset=steps_offset, prediction_type=prediction_type)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._ac = self.alphas_cumprod.double().numpy()

    # -- checkpoint format: <path>/scheduler/scheduler_config.json (test_fatezero.py:112-115) ------------------
    @classmethod
    def from_config(cls, config: dict):
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "set_alpha_to_one", "steps_offset",
                "prediction_type")
        kw = {k: config[k] for k in keys if k in config}
        # clip_sample / steps_offset are forced by the pipeline anyway (stable_diffusion.py:56-81); PNDM-style configs of the
        # SD checkpoints carry `skip_prk_steps` etc., which DDIM ignores like diffusers' from_config does
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **unused):
        import json
        import os
        root = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(root, "scheduler_config.json")) as f:
            return cls.from_config(json.load(f))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.config.steps_offset  # kept on the host: they drive Python loops

    # -- scalar coefficients: x' = cz * x + ce * eps ------------------------------------------------------
    def _alpha(self, t):
        return float(self._ac[t]) if t >= 0 else float(self._ac[0] if not self.config.set_alpha_to_one else 1.0)

    def step_coefficients(self, timestep):
        """DDIMScheduler.step(eta=0): x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); x' = sqrt(a_p) x0 + sqrt(1-a_p) e."""
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t, a_p = self._alpha(t), self._alpha(prev)
        cz = (a_p / a_t) ** 0.5
        ce = (1 - a_p) ** 0.5 - cz * (1 - a_t) ** 0.5
        return cz, ce

    def inverse_step_coefficients(self, timestep):
        """next_clean2noise_step (p2p_ddim_spatial_temporal.py:150-161)."""
        t = int(timestep)
        cur = min(t - self.config.num_train_timesteps // self.num_inference_steps, 999)
        a_t, a_n = self._alpha(cur), self._alpha(t)
        cz = (a_n / a_t) ** 0.5
        ce = (1 - a_n) ** 0.5 - cz * (1 - a_t) ** 0.5
        return cz, ce

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict: bool = True):
        if eta != 0.0:
            raise NotImplementedError("the editing path always runs eta = 0")
        cz, ce = self.step_coefficients(timestep)
        prev = cz * sample + ce * model_output
        return SimpleNamespace(prev_sample=prev, pred_original_sample=None) if return_dict else (prev,)
