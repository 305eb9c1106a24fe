"""Mirror of modules/sd_samplers_common.py for the path: `SamplerData` (:13-31), `InterruptedException` (:186),
`TorchHijack` (:214-235), `Sampler` base (:238-364: callback_state, launch_sampling, initialize)."""
import inspect
from collections import namedtuple

import torch

from . import shared
from .. import k_diffusion  # noqa: F401
from ..k_diffusion import sampling as kd_sampling


class SamplerData(namedtuple("SamplerData", ["name", "constructor", "aliases", "options"])):
    def total_steps(self, steps):
        if self.options.get("second_order", False):
            steps = steps * 2
        return steps


class InterruptedException(BaseException):
    pass


def setup_img2img_steps(p, steps=None):
    """sd_samplers_common.py:24-33."""
    if getattr(shared.opts, "img2img_fix_steps", False) or steps is not None:
        requested_steps = (steps or p.steps)
        steps = int(requested_steps / min(p.denoising_strength, 0.999)) if p.denoising_strength > 0 else 0
        t_enc = requested_steps - 1
    else:
        steps = p.steps
        t_enc = int(min(p.denoising_strength, 0.999) * steps)
    return steps, t_enc


def images_tensor_to_samples(image, approximation=None, model=None):
    """sd_samplers_common.py:96-120 ("Full" VAE encode only): image [B,3,H,W] in [0, 1] -> latent, one image at a time as the
    reference does (each draws its own posterior noise from the CPU default generator, nn/vae.py:28)."""
    model = model if model is not None else shared.sd_model
    image = image.to(model.device, dtype=torch.float32) * 2 - 1
    if len(image) > 1:
        return torch.stack([model.encode_first_stage(torch.unsqueeze(img, 0))[0] for img in image])
    return model.encode_first_stage(image)


class TorchHijack:
    """Replaces `torch` inside k_diffusion.sampling so that randn_like draws from the per-image generators (p.rng):
    images generated in a batch equal images generated individually (and independent of multi-GPU sharding)."""

    def __init__(self, p):
        self.rng = p.rng

    def __getattr__(self, item):
        if item == "randn_like":
            return self.randn_like
        if hasattr(torch, item):
            return getattr(torch, item)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{item}'")

    def randn_like(self, x):
        return self.rng.next()


class Sampler:
    def __init__(self, funcname):
        self.funcname = funcname
        self.func = funcname
        self.extra_params = []
        self.stop_at = None
        self.eta = None
        self.config = None
        self.last_latent = None
        self.s_min_uncond = None
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = 0.0, 0.0, float("inf"), 1.0
        self.eta_option_field = "eta_ancestral"
        self.eta_default = 1.0
        self.p = None
        self.model_wrap_cfg = None
        self.sampler_extra_args = None
        self.options = {}

    def callback_state(self, d):
        step = d["i"]
        if self.stop_at is not None and step > self.stop_at:
            raise InterruptedException
        shared.state.sampling_step = step

    def launch_sampling(self, steps, func):
        self.model_wrap_cfg.steps = steps
        self.model_wrap_cfg.total_steps = self.config.total_steps(steps)
        shared.state.sampling_steps = steps
        shared.state.sampling_step = 0
        try:
            return func()
        except RecursionError:
            return self.last_latent
        except InterruptedException:
            return self.last_latent

    def create_noise_sampler(self, x, sigmas, p):
        """sd_samplers_common.py:343-351: one Brownian path per image, seeded by the image's seed, so results do not depend on batch size."""
        if getattr(shared.opts, "no_dpmpp_sde_batch_determinism", False):
            return None
        sigma_min, sigma_max = sigmas[sigmas > 0].min(), sigmas.max()
        it = getattr(p, "iteration", 0)
        current_iter_seeds = p.all_seeds[it * p.batch_size:(it + 1) * p.batch_size]
        return kd_sampling.BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=current_iter_seeds)

    def initialize(self, p):
        self.p = p
        self.model_wrap_cfg.p = p
        self.model_wrap_cfg.mask = getattr(p, "mask", None)
        self.model_wrap_cfg.nmask = getattr(p, "nmask", None)
        self.model_wrap_cfg.mask_noise_source = getattr(p, "mask_noise_source", None)
        self.model_wrap_cfg.step = 0
        self.eta = p.eta if getattr(p, "eta", None) is not None else getattr(shared.opts, self.eta_option_field, 0.0)
        self.s_min_uncond = getattr(p, "s_min_uncond", 0.0)
        kd_sampling.torch = TorchHijack(p)
        extra = {}
        params = inspect.signature(self.func).parameters
        for name in self.extra_params:
            if hasattr(p, name) and name in params:
                extra[name] = getattr(p, name)
        if "eta" in params:
            extra["eta"] = self.eta
        return extra
