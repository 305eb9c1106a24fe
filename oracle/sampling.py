"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the k-diffusion pieces on the path:
  k_diffusion/sampling.py:19-25 get_sigmas_karras; :53-60 get_ancestral_step; :120-137 sample_euler;
  :141-159 sample_euler_ancestral; :649-671 sample_dpmpp_2m;
  k_diffusion/external.py:62-67 ForgeScheduleLinker.get_sigmas;
  modules/sd_schedulers.py:10-15 (to_d without append_dims).
`noise_fn()` stands for the TorchHijack'd torch.randn_like (modules/sd_samplers_common.py:214-235):
every call returns the next per-image-seeded noise tensor.
"""
import torch


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    min_inv, max_inv = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return append_zero((max_inv + ramp * (min_inv - max_inv)) ** rho)


def get_sigmas_linker(predictor, n):
    t = torch.linspace(len(predictor.sigmas) - 1, 0, n)
    return append_zero(predictor.sigma(t))


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler(model, x, sigmas, noise_fn=None, callback=None):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        if noise_fn is not None:
            noise_fn()  # sampling.py:126 draws randn_like(x) every step even with s_churn=0 (advances p.rng)
        sigma_hat = sigmas[i]
        denoised = model(x, sigma_hat * s_in)
        d = (x - denoised) / sigma_hat
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        x = x + d * (sigmas[i + 1] - sigma_hat)
    return x


def sample_euler_ancestral(model, x, sigmas, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        d = (x - denoised) / sigmas[i]
        x = x + d * (sigma_down - sigmas[i])
        if sigmas[i + 1] > 0:
            x = x + noise_fn() * s_noise * sigma_up
    return x


def sample_dpmpp_2m(model, x, sigmas, callback=None):
    s_in = x.new_ones([x.shape[0]])
    t_fn = lambda s: s.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    old = None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old is None or sigmas[i + 1] == 0:
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised
        else:
            r = (t - t_fn(sigmas[i - 1])) / h
            dd = (1 + 1 / (2 * r)) * denoised - (1 / (2 * r)) * old
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * dd
        old = denoised
    return x


SAMPLERS = {"Euler": (sample_euler, None), "Euler a": (sample_euler_ancestral, None),
            "DPM++ 2M": (sample_dpmpp_2m, "karras")}  # sd_samplers_kdiffusion.py:14-34 default schedulers
