"""`CompVisSampler` -- mirror of modules/sd_samplers_timesteps.py (table :13-18, `CompVisTimestepsDenoiser` :27-35,
`CFGDenoiserTimesteps` :38-49, `CompVisSampler.get_timesteps/sample_img2img/sample` :52-153): samplers that run on discrete
timesteps and eps, fed by the CFGDenoiser's classic_ddim_eps_estimation mode."""
import inspect

import torch

from . import sd_samplers_common, sd_samplers_timesteps_impl, shared
from .sd_samplers_cfg_denoiser import CFGDenoiser
from ..backend.sampling.sampling_function import sampling_cleanup, sampling_prepare

samplers_timesteps = [
    ("DDIM", sd_samplers_timesteps_impl.ddim, ["ddim"], {}),
    ("DDIM CFG++", sd_samplers_timesteps_impl.ddim_cfgpp, ["ddim_cfgpp"], {}),
    ("PLMS", sd_samplers_timesteps_impl.plms, ["plms"], {}),
    ("UniPC", sd_samplers_timesteps_impl.unipc, ["unipc"], {}),
]

samplers_data_timesteps = [
    sd_samplers_common.SamplerData(label, lambda model, funcname=funcname: CompVisSampler(funcname, model), aliases, options)
    for label, funcname, aliases, options in samplers_timesteps
]


class CompVisTimestepsDenoiser:
    def __init__(self, model):
        self.inner_model = model
        self.alphas_cumprod = 1.0 / (model.forge_objects.unet.model.predictor.sigmas ** 2.0 + 1.0)
        model.alphas_cumprod = self.alphas_cumprod  # the reference stores it on sd_model (:30)


class CFGDenoiserTimesteps(CFGDenoiser):
    def __init__(self, sampler, sd_model):
        super().__init__(sampler)
        self.classic_ddim_eps_estimation = True
        self._sd_model = sd_model

    @property
    def inner_model(self):
        if self.model_wrap is None:
            self.model_wrap = CompVisTimestepsDenoiser(self._sd_model)
        return self.model_wrap


class CompVisSampler(sd_samplers_common.Sampler):
    def __init__(self, funcname, sd_model):
        super().__init__(funcname)
        self.eta_option_field = "eta_ddim"
        self.eta_infotext_field = "Eta DDIM"
        self.eta_default = 0.0
        self.model_wrap_cfg = CFGDenoiserTimesteps(self, sd_model)
        self.model_wrap = self.model_wrap_cfg.inner_model

    def get_timesteps(self, p, steps):
        discard = self.config is not None and self.config.options.get("discard_next_to_last_sigma", False)
        if shared.opts.always_discard_next_to_last_sigma and not discard:
            discard = True
        steps += 1 if discard else 0
        return torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)  # host: the loops read it as python ints

    def _extra_args(self, p, conditioning, unconditional_conditioning, image_conditioning):
        return {"cond": conditioning, "image_cond": image_conditioning, "uncond": unconditional_conditioning, "cond_scale": p.cfg_scale,
                "s_min_uncond": self.s_min_uncond}

    def sample_img2img(self, p, x, noise, conditioning, unconditional_conditioning, steps=None, image_conditioning=None):
        unet_patcher = self.model_wrap.inner_model.forge_objects.unet
        sampling_prepare(unet_patcher, x=x)
        steps, t_enc = sd_samplers_common.setup_img2img_steps(p, steps)
        timesteps = self.get_timesteps(p, steps)
        timesteps_sched = timesteps[:t_enc]
        acd = self.model_wrap.alphas_cumprod
        a = float(acd[int(timesteps[t_enc])])
        from .. import hipops as ops
        x = x.to(noise)
        xi = ops.lincomb([x, noise], [a ** 0.5, (1 - a) ** 0.5])  # :87 q_sample at timesteps[t_enc]
        extra_noise = getattr(shared.opts, "img2img_extra_noise", 0.0)
        if extra_noise > 0:
            xi = ops.lincomb([xi, noise], [1.0, extra_noise * a ** 0.5])
        extra_params_kwargs = self.initialize(p)
        parameters = inspect.signature(self.func).parameters
        if "timesteps" in parameters:
            extra_params_kwargs["timesteps"] = timesteps_sched
        if "is_img2img" in parameters:
            extra_params_kwargs["is_img2img"] = True
        self.model_wrap_cfg.init_latent = x
        self.last_latent = x
        self.sampler_extra_args = self._extra_args(p, conditioning, unconditional_conditioning, image_conditioning)
        samples = self.launch_sampling(t_enc + 1, lambda: self.func(self.model_wrap_cfg, xi, extra_args=self.sampler_extra_args,
                                                                    disable=getattr(p, "disable_progress", True),
                                                                    callback=self.callback_state, **extra_params_kwargs))
        sampling_cleanup(unet_patcher)
        return samples

    def sample(self, p, x, conditioning, unconditional_conditioning, steps=None, image_conditioning=None):
        unet_patcher = self.model_wrap.inner_model.forge_objects.unet
        sampling_prepare(unet_patcher, x=x)
        steps = steps or p.steps
        timesteps = self.get_timesteps(p, steps)
        extra_params_kwargs = self.initialize(p)
        parameters = inspect.signature(self.func).parameters
        if "timesteps" in parameters:
            extra_params_kwargs["timesteps"] = timesteps
        self.last_latent = x
        self.sampler_extra_args = self._extra_args(p, conditioning, unconditional_conditioning, image_conditioning)
        samples = self.launch_sampling(steps, lambda: self.func(self.model_wrap_cfg, x, extra_args=self.sampler_extra_args,
                                                                disable=getattr(p, "disable_progress", True),
                                                                callback=self.callback_state, **extra_params_kwargs))
        sampling_cleanup(unet_patcher)
        return samples


VanillaStableDiffusionSampler = CompVisSampler
