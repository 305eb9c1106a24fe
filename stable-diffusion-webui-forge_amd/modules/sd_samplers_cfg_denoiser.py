"""`CFGDenoiser` -- mirror of modules/sd_samplers_cfg_denoiser.py:33-228 (forward(x, sigma, uncond, cond, cond_scale,
s_min_uncond, image_cond)).  Per step: interrupt check, prompt-schedule reconstruction, NGMS / skip-early-cond rules,
`sampling_function`, last-latent bookkeeping, inpaint latent-mask blending (:178-181 before the model, :204-213 after)."""
import torch

from . import prompt_parser, sd_samplers_common, shared
from .. import hipops as ops
from ..backend.modules.k_model import SigmaInfo
from ..backend.sampling.sampling_function import sampling_function


class CFGDenoiserParams:
    def __init__(self, x, image_cond, sigma, sampling_step, total_sampling_steps, text_cond, text_uncond, denoiser=None):
        self.x, self.image_cond, self.sigma = x, image_cond, sigma
        self.sampling_step, self.total_sampling_steps = sampling_step, total_sampling_steps
        self.text_cond, self.text_uncond, self.denoiser = text_cond, text_uncond, denoiser


class CFGDenoiser:
    def __init__(self, sampler):
        self.model_wrap = None
        self.mask = None
        self.nmask = None
        self.init_latent = None
        self.steps = None
        self.total_steps = None
        self.step = 0
        self.image_cfg_scale = None
        self.padded_cond_uncond = False
        self.padded_cond_uncond_v0 = False
        self.sampler = sampler
        self.p = None
        self.need_last_noise_uncond = False
        self.last_noise_uncond = None
        self.classic_ddim_eps_estimation = False
        # test hook: callable(step, like) replacing the torch.randn_like of :180 (device RNG in the reference, so fixtures
        # made on CPU can only be matched with an injected noise source)
        self.mask_noise_source = None

    @property
    def inner_model(self):
        raise NotImplementedError()

    def forward(self, x, sigma, uncond, cond, cond_scale, s_min_uncond, image_cond):
        state, opts = shared.state, shared.opts
        if state.interrupted or state.skipped:
            raise sd_samplers_common.InterruptedException
        sig0 = sigma.fmx_sigma.host[0] if hasattr(sigma, "fmx_sigma") else float(sigma[0])
        if self.classic_ddim_eps_estimation:
            # :163-169  `sigma` holds a timestep and x the variance-preserving latent: look the real sigma up and rescale x to the
            # variance-exploding latent the denoiser works on
            acd = self.inner_model.inner_model.alphas_cumprod
            fake_sigmas = ((1 - acd) / acd) ** 0.5
            sig0 = float(fake_sigmas[min(max(int(round(sig0)), 0), int(fake_sigmas.shape[0]) - 1)])
            x = ops.scale_f32(x, (sig0 ** 2.0 + 1.0) ** 0.5)
            sigma = torch.full_like(sigma, sig0)
            sigma.fmx_sigma = SigmaInfo([sig0] * x.shape[0])
        if self.mask is not None:
            # :178-181  x = x * nmask + noise_scaling(sigma, randn_like(init_latent), init_latent) * mask
            noise = self.mask_noise_source(self.step, self.init_latent) if self.mask_noise_source is not None else torch.randn_like(self.init_latent)
            predictor = self.inner_model.inner_model.forge_objects.unet.model.predictor
            noisy_initial_latent = predictor.noise_scaling(sig0, noise.to(self.init_latent), self.init_latent, max_denoise=False)
            x = ops.blend_masked(x, self._nmask32(x), noisy_initial_latent, self._mask32(x))
        cond_composition, cond = prompt_parser.reconstruct_multicond_batch(cond, self.step)
        uncond = prompt_parser.reconstruct_cond_batch(uncond, self.step) if uncond is not None else None
        denoiser_params = CFGDenoiserParams(x, image_cond, sigma, state.sampling_step, state.sampling_steps, cond, uncond, self)
        if getattr(self.p, "is_hr_pass", False):
            cond_scale = self.p.hr_cfg
        if opts.skip_early_cond > 0 and self.step / self.total_steps <= opts.skip_early_cond:
            cond_scale = 1.0
        elif (self.step % 2 or opts.s_min_uncond_all) and s_min_uncond > 0 and sig0 < s_min_uncond:
            cond_scale = 1.0
        denoised, cond_pred, uncond_pred = sampling_function(self, denoiser_params=denoiser_params, cond_scale=cond_scale,
                                                             cond_composition=cond_composition)
        if self.need_last_noise_uncond:
            self.last_noise_uncond = ops.lincomb([x, uncond_pred], [1.0 / sig0, -1.0 / sig0])  # :201-202
        if self.mask is not None:
            denoised = ops.blend_masked(denoised, self._nmask32(x), self.init_latent, self._mask32(x))  # :204-213
        self.sampler.last_latent = denoised
        state.current_latent = denoised
        self.step += 1
        if self.classic_ddim_eps_estimation:
            return ops.lincomb([x, denoised], [1.0 / sig0, -1.0 / sig0])  # :224-226 eps = (x - denoised) / sigma
        return denoised

    def _mask32(self, like):
        if getattr(self, "_m32_src", None) is not self.mask:
            self._m32_src = self.mask
            self._m32 = self.mask.to(device=like.device, dtype=torch.float32).expand_as(like).contiguous()
        return self._m32

    def _nmask32(self, like):
        if getattr(self, "_n32_src", None) is not self.nmask:
            self._n32_src = self.nmask
            self._n32 = self.nmask.to(device=like.device, dtype=torch.float32).expand_as(like).contiguous()
        return self._n32

    __call__ = forward
