"""`KDiffusionSampler` -- mirror of modules/sd_samplers_kdiffusion.py (sampler table :14-34, `get_sigmas` :81-134,
`sample` :196-244) over the native denoiser.  The whole table is built; the SDE family uses a native Brownian path instead of torchsde's."""
import inspect

import torch

from . import sd_samplers_common, sd_samplers_extra, sd_schedulers, shared
from .sd_samplers_cfg_denoiser import CFGDenoiser
from ..backend.sampling.sampling_function import sampling_cleanup, sampling_prepare
from ..k_diffusion import external as kd_external
from ..k_diffusion import sampling as kd_sampling

samplers_k_diffusion = [
    ("DPM++ 2M", "sample_dpmpp_2m", ["k_dpmpp_2m"], {"scheduler": "karras"}),
    ("DPM++ SDE", "sample_dpmpp_sde", ["k_dpmpp_sde"], {"scheduler": "karras", "second_order": True, "brownian_noise": True}),
    ("DPM++ 2M SDE", "sample_dpmpp_2m_sde", ["k_dpmpp_2m_sde"], {"scheduler": "exponential", "brownian_noise": True}),
    ("DPM++ 2M SDE Heun", "sample_dpmpp_2m_sde", ["k_dpmpp_2m_sde_heun"], {"scheduler": "exponential", "brownian_noise": True, "solver_type": "heun"}),
    ("DPM++ 2S a", "sample_dpmpp_2s_ancestral", ["k_dpmpp_2s_a"], {"scheduler": "karras", "uses_ensd": True, "second_order": True}),
    ("DPM++ 3M SDE", "sample_dpmpp_3m_sde", ["k_dpmpp_3m_sde"], {"scheduler": "exponential", "discard_next_to_last_sigma": True,
                                                                  "brownian_noise": True}),
    ("Euler a", "sample_euler_ancestral", ["k_euler_a", "k_euler_ancestral"], {"uses_ensd": True}),
    ("Euler", "sample_euler", ["k_euler"], {}),
    ("LMS", "sample_lms", ["k_lms"], {}),
    ("Heun", "sample_heun", ["k_heun"], {"second_order": True}),
    ("DPM2", "sample_dpm_2", ["k_dpm_2"], {"scheduler": "karras", "discard_next_to_last_sigma": True, "second_order": True}),
    ("DPM2 a", "sample_dpm_2_ancestral", ["k_dpm_2_a"], {"scheduler": "karras", "discard_next_to_last_sigma": True, "uses_ensd": True,
                                                         "second_order": True}),
    ("DPM fast", "sample_dpm_fast", ["k_dpm_fast"], {"uses_ensd": True}),
    ("DPM adaptive", "sample_dpm_adaptive", ["k_dpm_ad"], {"uses_ensd": True}),
    ("Restart", sd_samplers_extra.restart_sampler, ["restart"], {"scheduler": "karras", "second_order": True}),
    ("HeunPP2", "sample_heunpp2", ["heunpp2"], {}),
    ("IPNDM", "sample_ipndm", ["ipndm"], {}),
    ("IPNDM_V", "sample_ipndm_v", ["ipndm_v"], {}),
    ("DEIS", "sample_deis", ["deis"], {}),
]
# the SDE family draws its noise from a native Brownian path (k_diffusion/sampling.py BatchedBrownianTree): the reference's comes from
# torchsde.BrownianTree, whose value stream is torchsde's own, so those four match the reference in distribution, not bit for bit

sampler_extra_params = {
    "sample_euler": ["s_churn", "s_tmin", "s_tmax", "s_noise"], "sample_heun": ["s_churn", "s_tmin", "s_tmax", "s_noise"],
    "sample_dpm_2": ["s_churn", "s_tmin", "s_tmax", "s_noise"], "sample_dpm_fast": ["s_noise"], "sample_dpm_2_ancestral": ["s_noise"],
    "sample_dpmpp_2s_ancestral": ["s_noise"], "sample_dpmpp_sde": ["s_noise"], "sample_dpmpp_2m_sde": ["s_noise"],
    "sample_dpmpp_3m_sde": ["s_noise"],
}

samplers_data_k_diffusion = [
    sd_samplers_common.SamplerData(label, lambda model, funcname=funcname: KDiffusionSampler(funcname, model), aliases, options)
    for label, funcname, aliases, options in samplers_k_diffusion
]
k_diffusion_samplers_map = {x.name: x for x in samplers_data_k_diffusion}


class CFGDenoiserKDiffusion(CFGDenoiser):
    def __init__(self, sampler, sd_model):
        super().__init__(sampler)
        self._sd_model = sd_model

    @property
    def inner_model(self):
        if self.model_wrap is None:
            self.model_wrap = kd_external.ForgeScheduleLinker(self._sd_model.forge_objects.unet.model.predictor)
            self.model_wrap.inner_model = self._sd_model
        return self.model_wrap


class KDiffusionSampler(sd_samplers_common.Sampler):
    def __init__(self, funcname, sd_model, options=None):
        super().__init__(funcname)
        self.extra_params = sampler_extra_params.get(funcname, [])
        self.options = options or {}
        self.func = funcname if callable(funcname) else getattr(kd_sampling, self.funcname)
        self.model_wrap_cfg = CFGDenoiserKDiffusion(self, sd_model)
        self.model_wrap = self.model_wrap_cfg.inner_model

    def get_sigmas(self, p, steps):
        opts = shared.opts
        discard = self.config is not None and self.config.options.get("discard_next_to_last_sigma", False)
        if opts.always_discard_next_to_last_sigma and not discard:
            discard = True
        steps += 1 if discard else 0
        scheduler_name = (getattr(p, "hr_scheduler", None) if getattr(p, "is_hr_pass", False) else getattr(p, "scheduler", None)) or "Automatic"
        if scheduler_name == "Automatic":
            scheduler_name = self.config.options.get("scheduler", None)
        scheduler = sd_schedulers.schedulers_map.get(scheduler_name)
        m_min, m_max = self.model_wrap.sigmas[0].item(), self.model_wrap.sigmas[-1].item()
        sigma_min, sigma_max = (0.1, 10) if opts.use_old_karras_scheduler_sigmas else (m_min, m_max)
        if getattr(p, "sampler_noise_scheduler_override", None):
            sigmas = p.sampler_noise_scheduler_override(steps)
        elif scheduler is None or scheduler.function is None:
            sigmas = self.model_wrap.get_sigmas(steps)
        else:
            kw = {"sigma_min": sigma_min, "sigma_max": sigma_max}
            if opts.sigma_min != 0 and opts.sigma_min != m_min:
                kw["sigma_min"] = opts.sigma_min
            if opts.sigma_max != 0 and opts.sigma_max != m_max:
                kw["sigma_max"] = opts.sigma_max
            if scheduler.default_rho != -1 and opts.rho != 0 and opts.rho != scheduler.default_rho:
                kw["rho"] = opts.rho
            if scheduler.need_inner_model:
                kw["inner_model"] = self.model_wrap
            sigmas = scheduler.function(n=steps, **kw, device="cpu")
        if discard:
            sigmas = torch.cat([sigmas[:-2], sigmas[-1:]])
        return sigmas.cpu()

    def sample_img2img(self, p, x, noise, conditioning, unconditional_conditioning, steps=None, image_conditioning=None):
        """sd_samplers_kdiffusion.py:136-194: x = init latent, noise = p.rng.next(); runs the last t_enc + 1 sigmas."""
        unet_patcher = self.model_wrap.inner_model.forge_objects.unet
        sampling_prepare(unet_patcher, x=x)
        steps, t_enc = sd_samplers_common.setup_img2img_steps(p, steps)
        sigmas = self.get_sigmas(p, steps)
        sigma_sched = sigmas[steps - t_enc - 1:]
        x = x.to(noise)
        xi = self.model_wrap.predictor.noise_scaling(sigma_sched[0], noise, x, max_denoise=False)
        extra_noise = getattr(shared.opts, "img2img_extra_noise", 0.0)
        if extra_noise > 0:
            xi += noise * extra_noise
        extra_params_kwargs = self.initialize(p)
        parameters = inspect.signature(self.func).parameters
        if "sigma_min" in parameters:
            extra_params_kwargs["sigma_min"] = sigma_sched[-2]
        if "sigma_max" in parameters:
            extra_params_kwargs["sigma_max"] = sigma_sched[0]
        if "n" in parameters:
            extra_params_kwargs["n"] = len(sigma_sched) - 1
        if "sigma_sched" in parameters:
            extra_params_kwargs["sigma_sched"] = sigma_sched
        if "sigmas" in parameters:
            extra_params_kwargs["sigmas"] = sigma_sched
        if self.config.options.get("brownian_noise", False):
            extra_params_kwargs["noise_sampler"] = self.create_noise_sampler(x, sigmas, p)
        if self.config.options.get("solver_type", None) == "heun":
            extra_params_kwargs["solver_type"] = "heun"
        self.model_wrap_cfg.init_latent = x
        self.last_latent = x
        self.sampler_extra_args = {"cond": conditioning, "image_cond": image_conditioning, "uncond": unconditional_conditioning,
                                   "cond_scale": p.cfg_scale, "s_min_uncond": self.s_min_uncond}
        samples = self.launch_sampling(t_enc + 1, lambda: self.func(self.model_wrap_cfg, xi, extra_args=self.sampler_extra_args,
                                                                    disable=getattr(p, "disable_progress", True),
                                                                    callback=self.callback_state, **extra_params_kwargs))
        sampling_cleanup(unet_patcher)
        return samples

    def sample(self, p, x, conditioning, unconditional_conditioning, steps=None, image_conditioning=None):
        unet_patcher = self.model_wrap.inner_model.forge_objects.unet
        sampling_prepare(unet_patcher, x=x)
        steps = steps or p.steps
        sigmas = self.get_sigmas(p, steps)  # stays on the host: the native loops read it as python floats
        x = self.model_wrap.predictor.noise_scaling(sigmas[0], x, None, max_denoise=shared.opts.sgm_noise_multiplier)
        extra_params_kwargs = self.initialize(p)
        parameters = inspect.signature(self.func).parameters
        if "n" in parameters:
            extra_params_kwargs["n"] = steps
        if "sigma_min" in parameters:
            extra_params_kwargs["sigma_min"] = self.model_wrap.sigmas[0].item()
            extra_params_kwargs["sigma_max"] = self.model_wrap.sigmas[-1].item()
        if "sigmas" in parameters:
            extra_params_kwargs["sigmas"] = sigmas
        if self.config.options.get("brownian_noise", False):
            extra_params_kwargs["noise_sampler"] = self.create_noise_sampler(x, sigmas, p)
        if self.config.options.get("solver_type", None) == "heun":
            extra_params_kwargs["solver_type"] = "heun"
        self.last_latent = x
        self.sampler_extra_args = {"cond": conditioning, "image_cond": image_conditioning, "uncond": unconditional_conditioning,
                                   "cond_scale": p.cfg_scale, "s_min_uncond": self.s_min_uncond}
        samples = self.launch_sampling(steps, lambda: self.func(self.model_wrap_cfg, x, extra_args=self.sampler_extra_args,
                                                                disable=getattr(p, "disable_progress", True),
                                                                callback=self.callback_state, **extra_params_kwargs))
        sampling_cleanup(unet_patcher)
        return samples
