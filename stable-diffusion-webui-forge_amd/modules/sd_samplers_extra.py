"""`restart_sampler` -- mirror of modules/sd_samplers_extra.py:7-74 (Restart Sampling, Xu et al. 2023): Heun steps along a Karras
schedule, jumping back up to sigma ~2 from sigma ~0.1 (re-noising by the variance difference) `restart_times` times."""
import torch
import tqdm

from .. import hipops as ops
from ..k_diffusion import sampling as kd_sampling


@torch.no_grad()
def restart_sampler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_noise=1.0, restart_list=None):
    """restart_list: {min_sigma: [restart_steps, restart_times, max_sigma]}; None = the reference's automatic choice by step count."""
    extra_args = {} if extra_args is None else extra_args
    sigmas = sigmas.detach().float().cpu()
    steps = sigmas.shape[0] - 1
    if restart_list is None:
        restart_list = {}
        if steps >= 20:
            restart_steps, restart_times = (steps // 4, 2) if steps >= 36 else (9, 1)
            sigmas = kd_sampling.get_sigmas_karras(steps - restart_steps * restart_times, sigmas[-2].item(), sigmas[0].item())
            restart_list = {0.1: [restart_steps + 1, restart_times, 2]}
    nearest = lambda v: int(torch.argmin(abs(sigmas - v), dim=0))
    restart_at = {nearest(k): v for k, v in restart_list.items()}
    pairs = []
    for i in range(len(sigmas) - 1):
        pairs.append((sigmas[i], sigmas[i + 1]))
        if i + 1 in restart_at:
            r_steps, r_times, r_max = restart_at[i + 1]
            hi = nearest(r_max)
            if hi < i + 1:
                seg = kd_sampling.get_sigmas_karras(r_steps, sigmas[i + 1].item(), sigmas[hi].item())[:-1]
                for _ in range(r_times):
                    pairs.extend(zip(seg[:-1], seg[1:]))
    last = None
    for step_id, (old, new) in enumerate(tqdm.tqdm(pairs, disable=disable)):
        so, sn = float(old), float(new)
        if last is not None and last < old:
            x = ops.lincomb([x, kd_sampling.torch.randn_like(x)], [1.0, s_noise * float((old ** 2 - last ** 2) ** 0.5)])
        denoised = model(x, kd_sampling._sigma_vec(x, so), **extra_args)
        if callback is not None:
            callback({"x": x, "i": step_id, "sigma": new, "sigma_hat": old, "denoised": denoised})
        if sn == 0:
            x = ops.euler_step(x, denoised, so, sn)
        else:
            x = kd_sampling._heun_update(model, x, denoised, so, sn, extra_args)
        last = new
    return x
