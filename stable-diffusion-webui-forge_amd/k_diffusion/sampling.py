"""k-diffusion sampler loops of the path -- mirror of k_diffusion/sampling.py (`sample_euler` :120-137,
`sample_euler_ancestral` :141-159, `sample_heun` :189-214, `sample_dpm_2` :218-246, `sample_dpm_2_ancestral` :249-276,
`sample_lms` :325-341, `sample_dpmpp_2s_ancestral` :573-603, `sample_dpmpp_2m` :649-671, `sample_heunpp2` :771-823,
`sample_ipndm` :829-865, `sample_ipndm_v` :869-929, `sample_deis` :933-981, `get_sigmas_*` :19-38, `get_ancestral_step` :53-60),
same signatures: fn(model, x, sigmas, extra_args=None, callback=None, disable=None, ...) -> x.

Every update of these solvers is a linear combination of latent-sized fp32 tensors (x, denoised, stage values, derivative
history, noise) with coefficients that depend only on the schedule, so each one is ONE fused pass (fmx_sampler_lincomb /
fmx_sampler_euler_step) with host-computed coefficients instead of the reference's chain of 4-10 elementwise kernels.

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


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0, device="cpu"):
    ramp = torch.linspace(1, 0, n, device=device) ** rho
    sigmas = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
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


def _churn(x, st, sh, i, n, s_churn, s_tmin, s_tmax, s_noise):
    """Karras et al. Algorithm 2 preamble shared by Euler / Heun / DPM2 / HeunPP2 (sampling.py:125-129, 194-199): the noise tensor is
    drawn every step (advances p.rng) whether or not it is used."""
    gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sh[i] <= s_tmax else 0.0
    eps = torch.randn_like(x)
    sigma_hat = float(st[i] * (gamma + 1))
    if gamma > 0:
        x = ops.lincomb([x, eps], [1.0, s_noise * (sigma_hat ** 2 - sh[i] ** 2) ** 0.5])
    return x, sigma_hat


def _heun_update(model, x, denoised, sigma, sigma_next, extra_args, w1=0.5, w2=0.5):
    """x + dt * (w1 d + w2 d_2), d = (x - denoised)/sigma, d_2 = (x_2 - denoised_2)/sigma_next at the Euler predictor x_2."""
    dt = sigma_next - sigma
    x_2 = ops.euler_step(x, denoised, sigma, sigma_next)
    denoised_2 = model(x_2, _sigma_vec(x, sigma_next), **extra_args)
    k1, k2 = w1 * dt / sigma, w2 * dt / sigma_next
    return ops.lincomb([x, denoised, x_2, denoised_2], [1.0 + k1, -k1, k2, -k2])


@torch.no_grad()
def sample_heun(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    n = len(sh) - 1
    for i in trange(n, disable=disable):
        x, sigma_hat = _churn(x, st, sh, i, n, s_churn, s_tmin, s_tmax, s_noise)
        denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": sigma_hat, "denoised": denoised})
        if sh[i + 1] == 0:
            x = ops.euler_step(x, denoised, sigma_hat, sh[i + 1])
        else:
            x = _heun_update(model, x, denoised, sigma_hat, sh[i + 1], extra_args)
    return x


def _log_midpoint(a, b):
    return float(a.log().lerp(b.log(), 0.5).exp())  # fp32 tensor arithmetic, as sampling.py:240


def _dpm2_update(model, x, denoised, sigma, sigma_t, sigma_target, extra_args, noise=None, noise_scale=0.0):
    """DPM-Solver-2 stage pair (sampling.py:238-246, 269-275): midpoint in log sigma, then x + d_2 * (sigma_target - sigma)."""
    sigma_mid = _log_midpoint(sigma_t, torch.as_tensor(sigma_target, dtype=torch.float32))
    x_2 = ops.euler_step(x, denoised, sigma, sigma_mid)
    denoised_2 = model(x_2, _sigma_vec(x, sigma_mid), **extra_args)
    k = (float(sigma_target) - sigma) / sigma_mid
    if noise is None:
        return ops.lincomb([x, x_2, denoised_2], [1.0, k, -k])
    return ops.lincomb([x, x_2, denoised_2, noise], [1.0, k, -k, noise_scale])


@torch.no_grad()
def sample_dpm_2(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    n = len(sh) - 1
    for i in trange(n, disable=disable):
        x, sigma_hat = _churn(x, st, sh, i, n, s_churn, s_tmin, s_tmax, s_noise)
        denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": sigma_hat, "denoised": denoised})
        if sh[i + 1] == 0:
            x = ops.euler_step(x, denoised, sigma_hat, sh[i + 1])
        else:
            x = _dpm2_update(model, x, denoised, sigma_hat, torch.tensor(sigma_hat, dtype=torch.float32), st[i + 1], extra_args)
    return x


@torch.no_grad()
def sample_dpm_2_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    st, sh = _host(sigmas)
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(st[i], st[i + 1], eta=eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        if float(sigma_down) == 0:
            x = ops.euler_step(x, denoised, sh[i], 0.0)
        else:
            # the reference draws the noise after the second model call (sampling.py:273-275); the model call draws nothing, so
            # fetching it first leaves p.rng's sequence unchanged and lets the update be one fused pass
            x = _dpm2_update(model, x, denoised, sh[i], st[i], sigma_down, extra_args, noise=noise_sampler(st[i], st[i + 1]),
                             noise_scale=s_noise * float(sigma_up))
    return x


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None):
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    st, sh = _host(sigmas)
    t_fn = lambda s: s.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(st[i], st[i + 1], eta=eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        noise = noise_sampler(st[i], st[i + 1]) if sh[i + 1] > 0 else None
        tail = ([noise], [s_noise * float(sigma_up)]) if noise is not None else ([], [])
        if float(sigma_down) == 0:
            k = (0.0 - sh[i]) / sh[i]
            x = ops.lincomb([x, denoised] + tail[0], [1.0 + k, -k] + tail[1])
        else:
            t, t_next = t_fn(st[i]), t_fn(torch.as_tensor(sigma_down, dtype=torch.float32))
            h = t_next - t
            s_mid = t + 0.5 * h
            x_2 = ops.lincomb([x, denoised], [float(sigma_fn(s_mid) / sigma_fn(t)), float(-(-h * 0.5).expm1())])
            denoised_2 = model(x_2, _sigma_vec(x, float(sigma_fn(s_mid))), **extra_args)
            x = ops.lincomb([x, denoised_2] + tail[0], [float(sigma_fn(t_next) / sigma_fn(t)), float(-(-h).expm1())] + tail[1])
    return x


def linear_multistep_coeff(order, t, i, j):
    """Integral over [t_i, t_i+1] of the j-th Lagrange basis polynomial on nodes t_i ... t_(i-order+1) (sampling.py:311-321)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        prod = 1.0
        for k in range(order):
            if k != j:
                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


def _derivative(x, denoised, sigma):
    return ops.lincomb([x, denoised], [1.0 / sigma, -1.0 / sigma])  # to_d


def _multistep(model, x, sigmas, extra_args, callback, disable, keep, coeff_fn):
    """Shared loop of the explicit linear multistep samplers: x += sum_k c_k d_(i-k) with d = (x - denoised)/sigma kept for `keep`
    steps; coeff_fn(i) -> [c_0 (current d), c_1 (previous), ...]."""
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    hist = []
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        coeffs = [float(c) for c in coeff_fn(i)]
        prev = hist[::-1][:len(coeffs) - 1]
        if keep > 0:
            d = _derivative(x, denoised, sh[i])
            x = ops.lincomb([x, d] + prev, [1.0] + coeffs)
            hist = (hist + [d])[-keep:]
        else:
            x = ops.lincomb([x, denoised], [1.0 + coeffs[0] / sh[i], -coeffs[0] / sh[i]])
    return x


@torch.no_grad()
def sample_lms(model, x, sigmas, extra_args=None, callback=None, disable=None, order=4):
    nodes = sigmas.detach().cpu().numpy()
    return _multistep(model, x, sigmas, extra_args, callback, disable, order - 1,
                      lambda i: [linear_multistep_coeff(min(i + 1, order), nodes, i, j) for j in range(min(i + 1, order))])


_ADAMS_BASHFORTH = {1: (1.0,), 2: (3 / 2, -1 / 2), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}


@torch.no_grad()
def sample_ipndm(model, x, sigmas, extra_args=None, callback=None, disable=None, max_order=4):
    st, _ = _host(sigmas)
    return _multistep(model, x, sigmas, extra_args, callback, disable, max_order - 1,
                      lambda i: [(st[i + 1] - st[i]) * c for c in _ADAMS_BASHFORTH[min(max_order, i + 1)]])


def _ipndm_v_weights(t, i, order):
    """Variable-step Adams-Bashforth weights on nodes t_i, t_(i-1), ... (sampling.py:891-921); t: fp32 host tensor."""
    h_n = t[i + 1] - t[i]
    if order == 1:
        return [1.0]
    h_1 = t[i] - t[i - 1]
    w = [(2 + h_n / h_1) / 2, -(h_n / h_1) / 2]
    if order == 2:
        return w
    h_2 = t[i - 1] - t[i - 2]
    a = (1 - h_n / (3 * (h_n + h_1)) * (h_n * (h_n + h_1)) / (h_1 * (h_1 + h_2))) / 2
    w = [w[0] + a, w[1] - (1 + h_1 / h_2) * a, a * h_1 / h_2]
    if order == 3:
        return w
    h_3 = t[i - 2] - t[i - 3]
    b = ((1 - h_n / (3 * (h_n + h_1))) / 2 + (1 - h_n / (2 * (h_n + h_1))) * h_n / (6 * (h_n + h_1 + h_2))) \
        * (h_n * (h_n + h_1) * (h_n + h_1 + h_2)) / (h_1 * (h_1 + h_2) * (h_1 + h_2 + h_3))
    q = h_1 * (h_1 + h_2) / (h_2 * (h_2 + h_3))
    return [w[0] + b, w[1] - (1 + h_1 / h_2 + q) * b, w[2] + (h_1 / h_2 + q * (1 + h_2 / h_3)) * b, -b * q * h_1 / h_2]


@torch.no_grad()
def sample_ipndm_v(model, x, sigmas, extra_args=None, callback=None, disable=None, max_order=4):
    st, _ = _host(sigmas)
    return _multistep(model, x, sigmas, extra_args, callback, disable, max_order - 1,
                      lambda i: [(st[i + 1] - st[i]) * c for c in _ipndm_v_weights(st, i, min(max_order, i + 1))])


@torch.no_grad()
def sample_deis(model, x, sigmas, extra_args=None, callback=None, disable=None, max_order=3, deis_mode="tab"):
    from . import deis
    st, sh = _host(sigmas)
    table = deis.get_deis_coeff_list(st, max_order, deis_mode=deis_mode)

    def coeffs(i):
        order = 1 if sh[i + 1] <= 0 else min(max_order, i + 1)
        return [st[i + 1] - st[i]] if order == 1 else table[i][:order]
    return _multistep(model, x, sigmas, extra_args, callback, disable, max_order - 1, coeffs)


@torch.no_grad()
def sample_heunpp2(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    extra_args = {} if extra_args is None else extra_args
    st, sh = _host(sigmas)
    n = len(sh) - 1
    s_end = sh[-1]
    for i in trange(n, disable=disable):
        x, sigma_hat = _churn(x, st, sh, i, n, s_churn, s_tmin, s_tmax, s_noise)
        denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": sigma_hat, "denoised": denoised})
        dt = sh[i + 1] - sigma_hat
        if sh[i + 1] == s_end:
            x = ops.euler_step(x, denoised, sigma_hat, sh[i + 1])
        elif sh[i + 2] == s_end:
            w2 = sh[i + 1] / (2 * sh[0])
            x = _heun_update(model, x, denoised, sigma_hat, sh[i + 1], extra_args, w1=1 - w2, w2=w2)
        else:
            # three-stage Heun++: x_2 at sigma_(i+1), x_3 at sigma_(i+2), weights from the sigmas (sampling.py:806-822)
            x_2 = ops.euler_step(x, denoised, sigma_hat, sh[i + 1])
            denoised_2 = model(x_2, _sigma_vec(x, sh[i + 1]), **extra_args)
            x_3 = ops.euler_step(x_2, denoised_2, sh[i + 1], sh[i + 2])
            denoised_3 = model(x_3, _sigma_vec(x, sh[i + 2]), **extra_args)
            w = 3 * sh[0]
            w2, w3 = sh[i + 1] / w, sh[i + 2] / w
            k1, k2, k3 = (1 - w2 - w3) * dt / sigma_hat, w2 * dt / sh[i + 1], w3 * dt / sh[i + 2]
            x = ops.lincomb([x, denoised, x_2, denoised_2, x_3, denoised_3], [1.0 + k1, -k1, k2, -k2, k3, -k3])
    return x
