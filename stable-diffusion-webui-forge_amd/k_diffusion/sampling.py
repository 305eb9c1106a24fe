"""k-diffusion sampler loops of the path -- mirror of k_diffusion/sampling.py (`sample_euler` :120-137,
`sample_euler_ancestral` :141-159, `sample_dpmpp_2m` :649-671, `get_sigmas_karras` :19-25, `get_ancestral_step` :53-60),
same signatures: fn(model, x, sigmas, extra_args=None, callback=None, disable=None, ...) -> x.

The sigma schedule is tiny and host-resident (python floats drive the scalar coefficients, so no device scalar is ever
synchronised, unlike `sigmas[i]` indexing of a device tensor in the reference); the latent-sized updates are single
fused HIP kernels (fmx_sampler_euler_step / fmx_sampler_lincomb3) on fp32 latents, as the reference keeps them.
`torch` in this module is replaceable by a TorchHijack exactly as in the reference (sd_samplers_common.py:305): the
Euler samplers draw `torch.randn_like(x)` once per step.
"""
import math

import torch
from tqdm.auto import trange

from .. import hipops as ops
from ..backend.modules.k_model import SigmaInfo


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas).to(device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device="cpu"):
    sigmas = torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp()
    return append_zero(sigmas)


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def _host(sigmas):
    """fp32 host copy of the schedule + the same values as python floats (exact fp32 -> double)."""
    s = sigmas.detach().float().cpu()
    return s, [float(v) for v in s.tolist()]


def _sigma_vec(x, value):
    """`sigma * s_in` of the reference: a device [B] vector, tagged with its host value (no sync later)."""
    v = torch.full((x.shape[0],), value, dtype=torch.float32, device=x.device)
    v.fmx_sigma = SigmaInfo([value] * x.shape[0])
    return v


@torch.no_grad()
def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    n = len(sh) - 1
    for i in trange(n, disable=disable):
        gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sh[i] <= s_tmax else 0.0
        eps = torch.randn_like(x)  # drawn every step, also when gamma == 0 (sampling.py:126): advances p.rng
        sigma_hat = float(st[i] * (gamma + 1))
        if gamma > 0:
            x = x + eps * (s_noise * (sigma_hat ** 2 - sh[i] ** 2) ** 0.5)
        denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": sigma_hat, "denoised": denoised})
        x = ops.euler_step(x, denoised, sigma_hat, sh[i + 1])  # d = (x - denoised)/sigma_hat ; x + d*dt
    return x


@torch.no_grad()
def sample_euler_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    st, sh = _host(sigmas)
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(st[i], st[i + 1], eta=eta)  # fp32 tensor math, as the reference
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        if sh[i + 1] > 0:
            noise = noise_sampler(st[i], st[i + 1])
            x = ops.euler_step(x, denoised, sh[i], float(sigma_down), noise=noise, noise_scale=s_noise * float(sigma_up))
        else:
            x = ops.euler_step(x, denoised, sh[i], float(sigma_down))
    return x


@torch.no_grad()
def sample_dpmpp_2m(model, x, sigmas, extra_args=None, callback=None, disable=None):
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    t_fn = lambda s: s.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    old_denoised = None
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        t, t_next = t_fn(st[i]), t_fn(st[i + 1])
        h = t_next - t
        a = float(sigma_fn(t_next) / sigma_fn(t))
        e = float(-(-h).expm1())
        if old_denoised is None or sh[i + 1] == 0:
            x = ops.lincomb3(x, denoised, None, a, e, 0.0)
        else:
            h_last = t - t_fn(st[i - 1])
            r = h_last / h
            c1, c2 = float(1 + 1 / (2 * r)), float(1 / (2 * r))
            x = ops.lincomb3(x, denoised, old_denoised, a, e * c1, -e * c2)
        old_denoised = denoised
    return x
