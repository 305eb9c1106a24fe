"""LCM sampler -- mirror of modules/sd_samplers_lcm.py: `LCMCompVisDenoiser` (:10-66, here only its schedule side -- sigmas of
the 50 LCM training timesteps, `get_sigmas`, `sigma_to_t`, `t_to_sigma`; its forward() is never reached in Forge, where
CFGDenoiser.forward goes through sampling_function), `sample_lcm` (:69-83), `CFGDenoiserLCM` / `LCMSampler` (:86-100)."""
import torch
from tqdm.auto import trange

from . import sd_samplers_common, sd_samplers_kdiffusion
from .sd_samplers_cfg_denoiser import CFGDenoiser
from .. import hipops as ops
from ..k_diffusion import sampling as kd_sampling


class LCMCompVisDenoiser:
    def __init__(self, model):
        timesteps, original_timesteps = 1000, 50  # LCM was distilled on every 20th timestep
        self.skip_steps = timesteps // original_timesteps
        self.inner_model = model
        self.predictor = model.forge_objects.unet.model.predictor
        alphas_cumprod = 1.0 / (self.predictor.sigmas ** 2.0 + 1.0)
        valid = torch.zeros(original_timesteps, dtype=torch.float32)
        for x in range(original_timesteps):
            valid[original_timesteps - 1 - x] = alphas_cumprod[timesteps - 1 - x * self.skip_steps]
        self.sigmas = ((1 - valid) / valid) ** 0.5  # k_diffusion/external.py:126 DiscreteEpsDDPMDenoiser
        self.log_sigmas = self.sigmas.log()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def get_sigmas(self, n=None):
        if n is None:
            return kd_sampling.append_zero(self.sigmas.flip(0))
        start, end = self.sigma_to_t(self.sigma_max), self.sigma_to_t(self.sigma_min)
        return kd_sampling.append_zero(self.t_to_sigma(torch.linspace(start, end, n)))

    def sigma_to_t(self, sigma, quantize=None):
        dists = sigma.log() - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape) * self.skip_steps + (self.skip_steps - 1)

    def t_to_sigma(self, timestep):
        t = torch.clamp(((timestep - (self.skip_steps - 1)) / self.skip_steps).float(), min=0, max=(len(self.sigmas) - 1))
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()  # external.py:116-120
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()


@torch.no_grad()
def sample_lcm(model, x, sigmas, extra_args=None, callback=None, disable=None, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = kd_sampling.default_noise_sampler(x) if noise_sampler is None else noise_sampler
    st, sh = kd_sampling._host(sigmas)
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, kd_sampling._sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        x = denoised
        if sh[i + 1] > 0:
            x = ops.lincomb([denoised, noise_sampler(st[i], st[i + 1])], [1.0, sh[i + 1]])
    return x


class CFGDenoiserLCM(CFGDenoiser):
    def __init__(self, sampler, sd_model):
        super().__init__(sampler)
        self._sd_model = sd_model

    @property
    def inner_model(self):
        if self.model_wrap is None:
            self.model_wrap = LCMCompVisDenoiser(self._sd_model)
        return self.model_wrap


class LCMSampler(sd_samplers_kdiffusion.KDiffusionSampler):
    def __init__(self, funcname, sd_model, options=None):
        super().__init__(funcname, sd_model, options)
        self.model_wrap_cfg = CFGDenoiserLCM(self, sd_model)
        self.model_wrap = self.model_wrap_cfg.inner_model


samplers_lcm = [("LCM", sample_lcm, ["k_lcm"], {})]
samplers_data_lcm = [
    sd_samplers_common.SamplerData(label, lambda model, funcname=funcname: LCMSampler(funcname, model), aliases, options)
    for label, funcname, aliases, options in samplers_lcm
]
