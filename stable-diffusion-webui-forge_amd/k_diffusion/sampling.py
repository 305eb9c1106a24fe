"""k-diffusion sampler loops of the path -- mirror of k_diffusion/sampling.py (`sample_euler` :120-137,
`sample_euler_ancestral` :141-159, `sample_heun` :189-214, `sample_dpm_2` :218-246, `sample_dpm_2_ancestral` :249-276,
`sample_lms` :325-341, `DPMSolver` / `sample_dpm_fast` / `sample_dpm_adaptive` :368-569, `sample_dpmpp_2s_ancestral` :573-603,
`sample_dpmpp_sde` :607-645, `sample_dpmpp_2m_sde` :675-717, `sample_dpmpp_3m_sde` :721-768, `BrownianTreeNoiseSampler` :67-117, `sample_dpmpp_2m` :649-671, `sample_heunpp2` :771-823,
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


# ---- Brownian noise for the SDE family ------------------------------------------------------------------------------------------------
class BatchedBrownianTree:
    """Per-image Brownian motions W(t) on [t0, t1] with W(t0) = 0, queried as increments W(b) - W(a) (sampling.py:67-91).

    The reference delegates to torchsde.BrownianTree (not in this image; its value stream is torchsde's own), so the VALUES here are not
    torchsde's -- the process is: a sample path refined on demand.  Known points are kept sorted; a new time inside a known interval is
    drawn from the Brownian bridge between its neighbours, one outside extends the path by an independent increment.  Every increment
    ever returned is consistent with one path per image, reproducible from (seed, query sequence).  Draws come from per-image CPU
    generators (seed + a fixed offset, so the path is independent of the initial latent noise drawn from `seed` itself), like
    modules/rng.py does for the 'CPU' noise source."""

    _SEED_OFFSET = 0x42524F57  # decorrelates from ImageRNG's generators, which are seeded with the bare image seeds

    def __init__(self, x, t0, t1, seed=None, **kwargs):
        t0, t1, self.sign = self.sort(float(t0), float(t1))
        if seed is None:
            seed = torch.randint(0, 2 ** 63 - 1, []).item()
        self.batched = True
        try:
            assert len(seed) == x.shape[0]
        except TypeError:
            seed = [seed]
            self.batched = False
        self.shape = tuple(x.shape[1:]) if self.batched else tuple(x.shape)
        self.device = x.device
        self.gens = [torch.Generator(device="cpu").manual_seed((int(s) + self._SEED_OFFSET) % (2 ** 63)) for s in seed]
        self.times = [t0, t1]
        self.values = [torch.zeros((len(seed),) + self.shape, dtype=torch.float32, device=self.device), None]
        self.values[1] = self._normal((t1 - t0) ** 0.5)

    @staticmethod
    def sort(a, b):
        return (a, b, 1) if a < b else (b, a, -1)

    def _normal(self, std):
        z = torch.stack([torch.randn(self.shape, generator=g, dtype=torch.float32) for g in self.gens])
        return z.to(self.device) * float(std)

    def _value(self, t):
        import bisect
        i = bisect.bisect_left(self.times, t)
        if i < len(self.times) and self.times[i] == t:
            return self.values[i]
        if i == 0:  # before the first known time: independent increment backwards
            w = ops.lincomb([self.values[0], self._normal((self.times[0] - t) ** 0.5)], [1.0, -1.0])
        elif i == len(self.times):
            w = ops.lincomb([self.values[-1], self._normal((t - self.times[-1]) ** 0.5)], [1.0, 1.0])
        else:  # Brownian bridge between the neighbours
            ta, tb = self.times[i - 1], self.times[i]
            f = (t - ta) / (tb - ta)
            w = ops.lincomb([self.values[i - 1], self.values[i], self._normal(((t - ta) * (tb - t) / (tb - ta)) ** 0.5)], [1.0 - f, f, 1.0])
        self.times.insert(i, t)
        self.values.insert(i, w)
        return w

    def __call__(self, t0, t1):
        t0, t1, sign = self.sort(float(t0), float(t1))
        w = ops.lincomb([self._value(t1), self._value(t0)], [float(self.sign * sign), -float(self.sign * sign)])
        return w if self.batched else w[0]


class BrownianTreeNoiseSampler:
    """sampling.py:94-117: noise_sampler(sigma, sigma_next) = (W(t_next) - W(t)) / sqrt(|t_next - t|), t = transform(sigma)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        self.transform = transform
        t0, t1 = self.transform(torch.as_tensor(sigma_min)), self.transform(torch.as_tensor(sigma_max))
        self.tree = BatchedBrownianTree(x, t0, t1, seed)

    def __call__(self, sigma, sigma_next):
        t0, t1 = self.transform(torch.as_tensor(sigma)), self.transform(torch.as_tensor(sigma_next))
        return ops.scale_f32(self.tree(t0, t1), 1.0 / float((t1 - t0).abs().sqrt()))


def _positive_range(st):
    return st[st > 0].min(), st.max()


@torch.no_grad()
def sample_dpmpp_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None, r=1 / 2):
    st, sh = _host(sigmas)
    sigma_min, sigma_max = _positive_range(st)
    noise_sampler = BrownianTreeNoiseSampler(x, sigma_min, sigma_max) if noise_sampler is None else noise_sampler
    extra_args = {} if extra_args is None else extra_args
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        if sh[i + 1] == 0:
            x = ops.euler_step(x, denoised, sh[i], 0.0)
            continue
        t, t_next = t_fn(st[i]), t_fn(st[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        # stage 1: ancestral DPM-Solver++ step to the intermediate time s
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(torch.as_tensor(sd, dtype=torch.float32))
        x_2 = ops.lincomb([x, denoised, noise_sampler(sigma_fn(t), sigma_fn(s))],
                          [float(sigma_fn(s_) / sigma_fn(t)), float(-(t - s_).expm1()), s_noise * float(su)])
        denoised_2 = model(x_2, _sigma_vec(x, float(sigma_fn(s))), **extra_args)
        # stage 2: full step with the blended data prediction
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(torch.as_tensor(sd, dtype=torch.float32))
        k = float(-(t - t_next_).expm1())
        x = ops.lincomb([x, denoised, denoised_2, noise_sampler(sigma_fn(t), sigma_fn(t_next))],
                        [float(sigma_fn(t_next_) / sigma_fn(t)), k * (1 - fac), k * fac, s_noise * float(su)])
    return x


@torch.no_grad()
def sample_dpmpp_2m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None,
                        solver_type="midpoint"):
    if solver_type not in {"heun", "midpoint"}:
        raise ValueError("solver_type must be 'heun' or 'midpoint'")
    st, sh = _host(sigmas)
    sigma_min, sigma_max = _positive_range(st)
    noise_sampler = BrownianTreeNoiseSampler(x, sigma_min, sigma_max) if noise_sampler is None else noise_sampler
    extra_args = {} if extra_args is None else extra_args
    old_denoised, h_last = None, None
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        if sh[i + 1] == 0:
            x = denoised
        else:
            t, s = -st[i].log(), -st[i + 1].log()
            h = s - t
            eta_h = eta * h
            phi = (-h - eta_h).expm1().neg()
            srcs, coefs = [x, denoised], [float(st[i + 1] / st[i] * (-eta_h).exp()), float(phi)]
            if old_denoised is not None:
                r = h_last / h
                w = float((phi / (-h - eta_h) + 1) * (1 / r)) if solver_type == "heun" else float(0.5 * phi * (1 / r))
                coefs[1] += w
                srcs.append(old_denoised), coefs.append(-w)
            if eta:
                srcs.append(noise_sampler(st[i], st[i + 1]))
                coefs.append(float(st[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise))
            x = ops.lincomb(srcs, coefs)
            h_last = h
        old_denoised = denoised
    return x


@torch.no_grad()
def sample_dpmpp_3m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None):
    st, sh = _host(sigmas)
    sigma_min, sigma_max = _positive_range(st)
    noise_sampler = BrownianTreeNoiseSampler(x, sigma_min, sigma_max) if noise_sampler is None else noise_sampler
    extra_args = {} if extra_args is None else extra_args
    denoised_1, denoised_2, h_1, h_2 = None, None, None, None
    for i in trange(len(sh) - 1, disable=disable):
        denoised = model(x, _sigma_vec(x, sh[i]), **extra_args)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": st[i], "sigma_hat": st[i], "denoised": denoised})
        if sh[i + 1] == 0:
            x = denoised
        else:
            t, s = -st[i].log(), -st[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            cx, c0, c1, c2 = float(torch.exp(-h_eta)), float((-h_eta).expm1().neg()), 0.0, 0.0
            if h_2 is not None:
                # third order: d1, d2 are divided differences of (denoised, denoised_1, denoised_2); written out per tensor
                r0, r1 = h_1 / h, h_2 / h
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                a = float(phi_2 * (1 + r0 / (r0 + r1)) - phi_3 / (r0 + r1))   # weight of d1_0 = (denoised - denoised_1) / r0
                b = float(-phi_2 * r0 / (r0 + r1) + phi_3 / (r0 + r1))        # weight of d1_1 = (denoised_1 - denoised_2) / r1
                c0 += a / float(r0)
                c1 += -a / float(r0) + b / float(r1)
                c2 += -b / float(r1)
            elif h_1 is not None:
                r = h_1 / h
                w = float((h_eta.neg().expm1() / h_eta + 1) / r)
                c0, c1 = c0 + w, c1 - w
            srcs, coefs = [x, denoised], [cx, c0]
            if denoised_1 is not None and c1 != 0.0:
                srcs.append(denoised_1), coefs.append(c1)
            if denoised_2 is not None and c2 != 0.0:
                srcs.append(denoised_2), coefs.append(c2)
            if eta:
                srcs.append(noise_sampler(st[i], st[i + 1]))
                coefs.append(float(st[i + 1] * (-2 * h * eta).expm1().neg().sqrt() * s_noise))
            x = ops.lincomb(srcs, coefs)
            h_1, h_2 = h, h_1
        denoised_1, denoised_2 = denoised, denoised_1
    return x


# ---- DPM-Solver (fast / adaptive) ----------------------------------------------------------------------------------------------------
class PIDStepSizeController:
    """Step-size control of DPM-Solver adaptive (behaviour of sampling.py:368-394): the proposed factor is a product of powers of the last three
    inverse error norms -- exponents from the P / I / D coefficients divided by the solver order -- squashed by 1 + atan(f - 1); a step is
    accepted when the factor reaches `accept_safety`, and only accepted steps enter the history.  `h` is the running step (host float / 0-dim)."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h, self.accept_safety, self.eps = h, accept_safety, eps
        self.exponents = ((pcoeff + icoeff + dcoeff) / order, -(pcoeff + 2 * dcoeff) / order, dcoeff / order)
        self.history = None   # inverse errors of [this proposal, last accepted, the accepted one before]

    def propose_step(self, error):
        inv = 1.0 / (float(error) + self.eps)
        self.history = [inv, inv, inv] if self.history is None else [inv] + self.history[1:]
        raw = 1.0
        for e, w in zip(self.history, self.exponents):
            raw *= e ** w
        factor = 1.0 + math.atan(raw - 1.0)
        accepted = factor >= self.accept_safety
        if accepted:
            self.history = [self.history[0], self.history[0], self.history[1]]
        self.h = self.h * factor
        return accepted


class DPMSolver:
    """DPM-Solver-1/2/3 single steps in t = -log sigma on eps = (x - denoised) / sigma (sampling.py:397-543).  t values are fp32 host
    tensors (0-dim), exactly the arithmetic of the reference; each stage value is one fused linear combination of (x, eps, eps_r1, ...)."""

    def __init__(self, model, extra_args=None, eps_callback=None, info_callback=None):
        self.model, self.extra_args = model, (extra_args or {})
        self.eps_callback, self.info_callback = eps_callback, info_callback

    @staticmethod
    def t(sigma):
        return sigma.log().neg()

    @staticmethod
    def sigma(t):
        return torch.exp(-t)

    def eps(self, eps_cache, key, x, t, *args, **kwargs):
        if key in eps_cache:
            return eps_cache[key], eps_cache
        sig = float(self.sigma(t))
        denoised = self.model(x, _sigma_vec(x, sig), *args, **self.extra_args, **kwargs)
        eps = ops.lincomb([x, denoised], [1.0 / sig, -1.0 / sig])
        if self.eps_callback is not None:
            self.eps_callback()
        return eps, {key: eps, **eps_cache}

    def dpm_solver_1_step(self, x, t, t_next, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, "eps", x, t)
        return ops.lincomb([x, eps], [1.0, float(-self.sigma(t_next) * h.expm1())]), eps_cache

    def dpm_solver_2_step(self, x, t, t_next, r1=1 / 2, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, "eps", x, t)
        s1 = t + r1 * h
        u1 = ops.lincomb([x, eps], [1.0, float(-self.sigma(s1) * (r1 * h).expm1())])
        eps_r1, eps_cache = self.eps(eps_cache, "eps_r1", u1, s1)
        a, b = float(self.sigma(t_next) * h.expm1()), float(self.sigma(t_next) / (2 * r1) * h.expm1())
        return ops.lincomb([x, eps, eps_r1], [1.0, -a + b, -b]), eps_cache

    def dpm_solver_3_step(self, x, t, t_next, r1=1 / 3, r2=2 / 3, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, "eps", x, t)
        s1, s2 = t + r1 * h, t + r2 * h
        u1 = ops.lincomb([x, eps], [1.0, float(-self.sigma(s1) * (r1 * h).expm1())])
        eps_r1, eps_cache = self.eps(eps_cache, "eps_r1", u1, s1)
        a = float(self.sigma(s2) * (r2 * h).expm1())
        b = float(self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1))
        u2 = ops.lincomb([x, eps, eps_r1], [1.0, -a + b, -b])
        eps_r2, eps_cache = self.eps(eps_cache, "eps_r2", u2, s2)
        a = float(self.sigma(t_next) * h.expm1())
        b = float(self.sigma(t_next) / r2 * (h.expm1() / h - 1))
        return ops.lincomb([x, eps, eps_r2], [1.0, -a + b, -b]), eps_cache

    def _ancestral_target(self, t, t_next, t_end, eta):
        if not eta:
            return t_next, 0.0
        sd, su = get_ancestral_step(self.sigma(t), self.sigma(t_next), eta)
        t_next_ = torch.minimum(t_end, self.t(torch.as_tensor(sd, dtype=torch.float32)))
        return t_next_, float((self.sigma(t_next) ** 2 - self.sigma(t_next_) ** 2) ** 0.5)

    def dpm_solver_fast(self, x, t_start, t_end, nfe, eta=0.0, s_noise=1.0, noise_sampler=None):
        noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
        if not t_end > t_start and eta:
            raise ValueError("eta must be 0 for reverse sampling")
        m = math.floor(nfe / 3) + 1
        ts = torch.linspace(t_start, t_end, m + 1)
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        for i in range(len(orders)):
            eps_cache = {}
            t, t_next = ts[i], ts[i + 1]
            t_next_, su = self._ancestral_target(t, t_next, t_end, eta)
            eps, eps_cache = self.eps(eps_cache, "eps", x, t)
            if self.info_callback is not None:
                denoised = ops.lincomb([x, eps], [1.0, float(-self.sigma(t))])
                self.info_callback({"x": x, "i": i, "t": ts[i], "t_up": t, "denoised": denoised})
            step = {1: self.dpm_solver_1_step, 2: self.dpm_solver_2_step, 3: self.dpm_solver_3_step}[orders[i]]
            x, eps_cache = step(x, t, t_next_, eps_cache=eps_cache)
            noise = noise_sampler(self.sigma(t), self.sigma(t_next))  # drawn every step, also at eta = 0 (:497)
            if su != 0.0:
                x = ops.lincomb([x, noise], [1.0, su * s_noise])
        return x

    def dpm_solver_adaptive(self, x, t_start, t_end, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0.0, icoeff=1.0, dcoeff=0.0,
                            accept_safety=0.81, eta=0.0, s_noise=1.0, noise_sampler=None):
        noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
        if order not in {2, 3}:
            raise ValueError("order should be 2 or 3")
        forward = t_end > t_start
        if not forward and eta:
            raise ValueError("eta must be 0 for reverse sampling")
        h_init = abs(h_init) * (1 if forward else -1)
        s = t_start
        x_prev = x
        pid = PIDStepSizeController(h_init, pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
        info = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
        while s < t_end - 1e-5 if forward else s > t_end + 1e-5:
            eps_cache = {}
            t = torch.minimum(t_end, s + pid.h) if forward else torch.maximum(t_end, s + pid.h)
            t_, su = self._ancestral_target(s, t, t_end, eta)
            eps, eps_cache = self.eps(eps_cache, "eps", x, s)
            if order == 2:
                x_low, eps_cache = self.dpm_solver_1_step(x, s, t_, eps_cache=eps_cache)
                x_high, eps_cache = self.dpm_solver_2_step(x, s, t_, eps_cache=eps_cache)
            else:
                x_low, eps_cache = self.dpm_solver_2_step(x, s, t_, r1=1 / 3, eps_cache=eps_cache)
                x_high, eps_cache = self.dpm_solver_3_step(x, s, t_, eps_cache=eps_cache)
            error = ops.error_norm(x_low, x_high, x_prev, atol, rtol)  # the one host sync per step the controller needs
            accept = pid.propose_step(error)
            denoised = ops.lincomb([x, eps], [1.0, float(-self.sigma(s))]) if self.info_callback is not None else None
            if accept:
                x_prev = x_low
                noise = noise_sampler(self.sigma(s), self.sigma(t))
                x = ops.lincomb([x_high, noise], [1.0, su * s_noise]) if su != 0.0 else x_high
                s = t
                info["n_accept"] += 1
            else:
                info["n_reject"] += 1
            info["nfe"] += order
            info["steps"] += 1
            if self.info_callback is not None:
                self.info_callback({"x": x, "i": info["steps"] - 1, "t": s, "t_up": s, "denoised": denoised, "error": error, "h": pid.h, **info})
        return x, info


@torch.no_grad()
def sample_dpm_fast(model, x, sigma_min, sigma_max, n, extra_args=None, callback=None, disable=None, eta=0.0, s_noise=1.0, noise_sampler=None):
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    from tqdm.auto import tqdm
    with tqdm(total=n, disable=disable) as pbar:
        dpm_solver = DPMSolver(model, extra_args, eps_callback=pbar.update)
        if callback is not None:
            dpm_solver.info_callback = lambda info: callback({"sigma": dpm_solver.sigma(info["t"]), "sigma_hat": dpm_solver.sigma(info["t_up"]), **info})
        return dpm_solver.dpm_solver_fast(x, dpm_solver.t(torch.tensor(float(sigma_max))), dpm_solver.t(torch.tensor(float(sigma_min))), n, eta,
                                          s_noise, noise_sampler)


@torch.no_grad()
def sample_dpm_adaptive(model, x, sigma_min, sigma_max, extra_args=None, callback=None, disable=None, order=3, rtol=0.05, atol=0.0078,
                        h_init=0.05, pcoeff=0.0, icoeff=1.0, dcoeff=0.0, accept_safety=0.81, eta=0.0, s_noise=1.0, noise_sampler=None,
                        return_info=False):
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    from tqdm.auto import tqdm
    with tqdm(disable=disable) as pbar:
        dpm_solver = DPMSolver(model, extra_args, eps_callback=pbar.update)
        if callback is not None:
            dpm_solver.info_callback = lambda info: callback({"sigma": dpm_solver.sigma(info["t"]), "sigma_hat": dpm_solver.sigma(info["t_up"]), **info})
        x, info = dpm_solver.dpm_solver_adaptive(x, dpm_solver.t(torch.tensor(float(sigma_max))), dpm_solver.t(torch.tensor(float(sigma_min))), order,
                                                 rtol, atol, h_init, pcoeff, icoeff, dcoeff, accept_safety, eta, s_noise, noise_sampler)
    if return_info:
        return x, info
    return x
