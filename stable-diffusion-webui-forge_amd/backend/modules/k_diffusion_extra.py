"""Samplers Forge adds on top of A1111's table -- mirror of backend/modules/k_diffusion_extra.py (`generic_step_sampler` :12-25,
`DDPMSampler_step` :28-37, `sample_ddpm` :40-42).  The ancestral DDPM step works on the variance-preserving latent
x / sqrt(1 + sigma^2); written out, the whole step (to VP, posterior mean, noise, back to VE) is linear in (x, denoised, noise)."""
import torch
from tqdm.auto import trange

from ... import hipops as ops
from ...k_diffusion import sampling as kd_sampling


def ddpm_step_coefficients(sigma, sigma_prev):
    """-> (coefficient of x, of denoised, of noise) for one step sigma -> sigma_prev; fp32 tensor arithmetic as the reference's."""
    alpha_cumprod = 1 / ((sigma * sigma) + 1)
    alpha_cumprod_prev = 1 / ((sigma_prev * sigma_prev) + 1)
    alpha = alpha_cumprod / alpha_cumprod_prev
    to_vp = 1 / torch.sqrt(1.0 + sigma ** 2.0)
    k_eps = (1 - alpha) / (1 - alpha_cumprod).sqrt()            # on noise_pred = (x - denoised) / sigma
    back = torch.sqrt(1.0 + sigma_prev ** 2.0) if sigma_prev != 0 else torch.tensor(1.0)
    pre = (1.0 / alpha).sqrt() * back
    cx = pre * (to_vp - k_eps / sigma)
    cd = pre * k_eps / sigma
    cn = ((1 - alpha) * (1.0 - alpha_cumprod_prev) / (1.0 - alpha_cumprod)).sqrt() * back if sigma_prev > 0 else torch.tensor(0.0)
    return float(cx), float(cd), float(cn)


@torch.no_grad()
def sample_ddpm(model, x, sigmas, extra_args=None, callback=None, disable=None, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = kd_sampling.default_noise_sampler(x) if noise_sampler is None else noise_sampler
    st, sh = kd_sampling._host(sigmas)
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, kd_sampling._sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        cx, cd, cn = ddpm_step_coefficients(st[i], st[i + 1])
        if sh[i + 1] > 0:
            x = ops.lincomb([x, denoised, noise_sampler(st[i], st[i + 1])], [cx, cd, cn])
        else:
            x = ops.lincomb([x, denoised], [cx, cd])
    return x
