"""DDIM / DDIM CFG++ / PLMS / UniPC over discrete timesteps -- mirror of modules/sd_samplers_timesteps_impl.py (`ddim` :11-42,
`ddim_cfgpp` :45-83, `plms` :86-142, `UniPCCFG` :145-170, `unipc` :173-181).  `model(x, t)` is a CFGDenoiser in classic_ddim_eps_estimation mode: x is the
variance-preserving latent, the return value is eps.  Every update is linear in (x, eps, eps history, noise) with coefficients
from the alphas_cumprod table, so it is one fused pass (fmx_sampler_lincomb) per step."""
import torch
import tqdm

from . import shared
from .models.diffusion.uni_pc import uni_pc
from .. import hipops as ops
from ..k_diffusion import sampling as kd_sampling
from ..backend.modules.k_model import SigmaInfo


def _tvec(x, t):
    """`t * s_in`: device [B] vector of the (integer-valued) timestep, tagged with its host value so nothing downstream syncs."""
    v = torch.full((x.shape[0],), float(t), dtype=torch.float32, device=x.device)
    v.fmx_sigma = SigmaInfo([float(t)] * x.shape[0])
    return v


def _tables(model, timesteps, eta=0.0):
    """Host float64 copies of the per-step constants (:12-16): alpha_t, alpha_prev, sqrt(1 - alpha_t), DDIM sigma_t."""
    acd = model.inner_model.inner_model.alphas_cumprod.detach().float().cpu()
    ts = torch.as_tensor(timesteps).long().cpu()
    alphas = acd[ts]
    alphas_prev = acd[torch.nn.functional.pad(ts[:-1], pad=(1, 0))].to(torch.float64)
    sqrt_one_minus_alphas = torch.sqrt(1 - alphas)
    a64 = alphas.to(torch.float64)
    sigmas = eta * torch.sqrt((1 - alphas_prev) / (1 - a64) * (1 - a64 / alphas_prev))
    return ts.tolist(), alphas.tolist(), alphas_prev.tolist(), sqrt_one_minus_alphas.tolist(), sigmas.tolist()


def _ddim_loop(model, x, timesteps, extra_args, callback, disable, eta, direction_from_uncond):
    ts, alphas, alphas_prev, s1m, sigmas = _tables(model, timesteps, eta)
    extra_args = {} if extra_args is None else extra_args
    for i in tqdm.trange(len(ts) - 1, disable=disable):
        index = len(ts) - 1 - i
        e_t = model(x, _tvec(x, ts[index]), **extra_args)
        direction = model.last_noise_uncond if direction_from_uncond else e_t
        a_t, a_prev, sigma_t = alphas[index], alphas_prev[index], sigmas[index]
        inv, k_dir = 1.0 / a_t ** 0.5, (1.0 - a_prev - sigma_t ** 2) ** 0.5
        noise = kd_sampling.torch.randn_like(x)  # drawn every step, also at eta = 0 (:36)
        pred_x0 = ops.lincomb([x, e_t], [inv, -s1m[index] * inv]) if callback is not None else None
        # x' = sqrt(a_prev) * (x - sqrt(1 - a_t) e_t) / sqrt(a_t) + sqrt(1 - a_prev - sigma_t^2) * direction + sigma_t * noise
        srcs, coefs = [x, e_t], [a_prev ** 0.5 * inv, -a_prev ** 0.5 * inv * s1m[index]]
        if direction is e_t:
            coefs[1] += k_dir
        else:
            srcs.append(direction), coefs.append(k_dir)
        if sigma_t != 0.0:
            srcs.append(noise), coefs.append(sigma_t)
        x = ops.lincomb(srcs, coefs)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": 0, "sigma_hat": 0, "denoised": pred_x0})
    return x


@torch.no_grad()
def ddim(model, x, timesteps, extra_args=None, callback=None, disable=None, eta=0.0):
    return _ddim_loop(model, x, timesteps, extra_args, callback, disable, eta, False)


@torch.no_grad()
def ddim_cfgpp(model, x, timesteps, extra_args=None, callback=None, disable=None, eta=0.0):
    """CFG++ (Chung et al. 2024): the re-noising direction uses the UNCONDITIONAL eps.  (The reference also sets
    `model.cond_scale_miltiplier = 1 / 12.5`, an attribute nothing in Forge reads, so cfg_scale is applied as is.)"""
    model.cond_scale_miltiplier = 1 / 12.5
    model.need_last_noise_uncond = True
    return _ddim_loop(model, x, timesteps, extra_args, callback, disable, eta, True)


_PLMS_WEIGHTS = {1: (3 / 2, -1 / 2), 2: (23 / 12, -16 / 12, 5 / 12), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}


@torch.no_grad()
def plms(model, x, timesteps, extra_args=None, callback=None, disable=None):
    ts, alphas, alphas_prev, s1m, _ = _tables(model, timesteps)
    extra_args = {} if extra_args is None else extra_args
    old_eps = []

    def step_coefs(index):
        # x_prev = sqrt(a_prev) (x - sqrt(1 - a_t) e) / sqrt(a_t) + sqrt(1 - a_prev) e  ->  (coefficient of x, coefficient of e)
        a_t, a_prev = alphas[index], alphas_prev[index]
        r = (a_prev / a_t) ** 0.5
        return r, (1.0 - a_prev) ** 0.5 - r * s1m[index]

    for i in tqdm.trange(len(ts) - 1, disable=disable):
        index = len(ts) - 1 - i
        e_t = model(x, _tvec(x, ts[index]), **extra_args)
        cx, ce = step_coefs(index)
        if len(old_eps) == 0:
            # pseudo improved Euler: eps at the Euler predictor, averaged (:119-123)
            x_euler = ops.lincomb([x, e_t], [cx, ce])
            e_next = model(x_euler, _tvec(x, ts[max(index - 1, 0)]), **extra_args)
            srcs, weights = [e_t, e_next], (0.5, 0.5)
        else:
            hist = old_eps[::-1]
            weights = _PLMS_WEIGHTS[len(hist)]  # Adams-Bashforth on eps, orders 2..4 (:124-132)
            srcs = [e_t] + hist
        inv = 1.0 / alphas[index] ** 0.5
        pred_x0 = ops.lincomb([x] + srcs, [inv] + [-s1m[index] * inv * w for w in weights]) if callback is not None else None
        x = ops.lincomb([x] + srcs, [cx] + [ce * w for w in weights])
        old_eps = (old_eps + [e_t])[-3:]
        if callback is not None:
            callback({"x": x, "i": i, "sigma": 0, "sigma_hat": 0, "denoised": pred_x0})
    return x


class UniPCCFG(uni_pc.UniPC):
    def __init__(self, cfg_model, extra_args, callback, *args, **kwargs):
        super().__init__(None, *args, **kwargs)

        def after_update(x, model_x):
            if callback is not None:
                callback({"x": x, "i": self.index, "sigma": 0, "sigma_hat": 0, "denoised": model_x})
            self.index += 1
        self.cfg_model = cfg_model
        self.extra_args = {} if extra_args is None else extra_args
        self.callback = callback
        self.index = 0
        self.after_update = after_update

    def get_model_input_time(self, t_continuous):
        return (t_continuous - 1.0 / self.noise_schedule.total_N) * 1000.0

    def model(self, x, t):
        return self.cfg_model(x, _tvec(x, float(self.get_model_input_time(t))), **self.extra_args)


@torch.no_grad()
def unipc(model, x, timesteps, extra_args=None, callback=None, disable=None, is_img2img=False):
    alphas_cumprod = model.inner_model.inner_model.alphas_cumprod
    ns = uni_pc.NoiseScheduleVP("discrete", alphas_cumprod=alphas_cumprod)
    t_start = float(timesteps[-1]) / 1000 + 1 / 1000 if is_img2img else None
    opts = shared.opts
    sampler = UniPCCFG(model, extra_args, callback, ns, predict_x0=True, thresholding=False, variant=opts.uni_pc_variant)
    return sampler.sample(x, steps=len(timesteps), t_start=t_start, skip_type=opts.uni_pc_skip_type, method="multistep", order=opts.uni_pc_order,
                          lower_order_final=opts.uni_pc_lower_order_final, disable=True if disable is None else disable)
