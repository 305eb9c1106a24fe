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


# ---- the rest of the k-diffusion table ("remaining samplers"; the torchsde-driven SDE family is excluded) -------
def _to_d(x, sigma, denoised):
    return (x - denoised) / sigma  # modules/sd_schedulers.py:10-12


def _cb(callback, x, i, sigma, sigma_hat, denoised):
    if callback is not None:
        callback({"x": x, "i": i, "sigma": sigma, "sigma_hat": sigma_hat, "denoised": denoised})


def sample_heun(model, x, sigmas, noise_fn=None, callback=None):
    """k_diffusion/sampling.py:189-214 with s_churn = 0 (gamma = 0): one randn_like per step is still drawn (:196)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        if noise_fn is not None:
            noise_fn()
        denoised = model(x, sigmas[i] * s_in)
        d = _to_d(x, sigmas[i], denoised)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        dt = sigmas[i + 1] - sigmas[i]
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            d_2 = _to_d(x_2, sigmas[i + 1], model(x_2, sigmas[i + 1] * s_in))
            x = x + (d + d_2) / 2 * dt
    return x


def sample_dpm_2(model, x, sigmas, noise_fn=None, callback=None):
    """k_diffusion/sampling.py:218-246, gamma = 0."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        if noise_fn is not None:
            noise_fn()
        denoised = model(x, sigmas[i] * s_in)
        d = _to_d(x, sigmas[i], denoised)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sigmas[i + 1] == 0:
            x = x + d * (sigmas[i + 1] - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigmas[i + 1].log(), 0.5).exp()
            x_2 = x + d * (sigma_mid - sigmas[i])
            d_2 = _to_d(x_2, sigma_mid, model(x_2, sigma_mid * s_in))
            x = x + d_2 * (sigmas[i + 1] - sigmas[i])
    return x


def sample_dpm_2_ancestral(model, x, sigmas, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    """k_diffusion/sampling.py:249-276."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        d = _to_d(x, sigmas[i], denoised)
        if sigma_down == 0:
            x = x + d * (sigma_down - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigma_down.log(), 0.5).exp()
            x_2 = x + d * (sigma_mid - sigmas[i])
            d_2 = _to_d(x_2, sigma_mid, model(x_2, sigma_mid * s_in))
            x = x + d_2 * (sigma_down - sigmas[i])
            x = x + noise_fn() * s_noise * sigma_up
    return x


def sample_dpmpp_2s_ancestral(model, x, sigmas, noise_fn, eta=1.0, s_noise=1.0, callback=None):
    """k_diffusion/sampling.py:573-603."""
    s_in = x.new_ones([x.shape[0]])
    t_fn = lambda s: s.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        if sigma_down == 0:
            x = x + _to_d(x, sigmas[i], denoised) * (sigma_down - sigmas[i])
        else:
            t, t_next = t_fn(sigmas[i]), t_fn(sigma_down)
            h = t_next - t
            s = t + 0.5 * h
            x_2 = (sigma_fn(s) / sigma_fn(t)) * x - (-h * 0.5).expm1() * denoised
            denoised_2 = model(x_2, sigma_fn(s) * s_in)
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised_2
        if sigmas[i + 1] > 0:
            x = x + noise_fn() * s_noise * sigma_up
    return x


def linear_multistep_coeff(order, t, i, j):
    """k_diffusion/sampling.py:311-321: integral over [t_i, t_i+1] of the j-th Lagrange basis on the last `order` nodes."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        p = 1.0
        for k in range(order):
            if k != j:
                p *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return p
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


def sample_lms(model, x, sigmas, order=4, callback=None):
    """k_diffusion/sampling.py:325-341."""
    s_in = x.new_ones([x.shape[0]])
    nodes = sigmas.detach().cpu().numpy()
    ds = []
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        ds.append(_to_d(x, sigmas[i], denoised))
        if len(ds) > order:
            ds.pop(0)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        cur = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur, nodes, i, j) for j in range(cur)]
        x = x + sum(c * d for c, d in zip(coeffs, reversed(ds)))
    return x


def sample_heunpp2(model, x, sigmas, noise_fn=None, callback=None):
    """k_diffusion/sampling.py:771-823, gamma = 0."""
    s_in = x.new_ones([x.shape[0]])
    s_end = sigmas[-1]
    for i in range(len(sigmas) - 1):
        if noise_fn is not None:
            noise_fn()
        denoised = model(x, sigmas[i] * s_in)
        d = _to_d(x, sigmas[i], denoised)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        dt = sigmas[i + 1] - sigmas[i]
        if sigmas[i + 1] == s_end:
            x = x + d * dt
            continue
        x_2 = x + d * dt
        d_2 = _to_d(x_2, sigmas[i + 1], model(x_2, sigmas[i + 1] * s_in))
        if sigmas[i + 2] == s_end:
            w2 = sigmas[i + 1] / (2 * sigmas[0])
            x = x + (d * (1 - w2) + d_2 * w2) * dt
        else:
            x_3 = x_2 + d_2 * (sigmas[i + 2] - sigmas[i + 1])
            d_3 = _to_d(x_3, sigmas[i + 2], model(x_3, sigmas[i + 2] * s_in))
            w = 3 * sigmas[0]
            w2, w3 = sigmas[i + 1] / w, sigmas[i + 2] / w
            x = x + ((1 - w2 - w3) * d + w2 * d_2 + w3 * d_3) * dt
    return x


_AB = {1: (1.0,), 2: (3 / 2, -1 / 2), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}


def sample_ipndm(model, x, sigmas, max_order=4, callback=None):
    """k_diffusion/sampling.py:829-865: Adams-Bashforth on d with fixed coefficients; history of max_order - 1 derivatives."""
    s_in = x.new_ones([x.shape[0]])
    hist = []
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        d = (x - denoised) / sigmas[i]
        order = min(max_order, i + 1)
        c = _AB[order]
        comb = c[0] * d
        for k in range(1, order):
            comb = comb + c[k] * hist[-k]
        x = x + (sigmas[i + 1] - sigmas[i]) * comb
        hist.append(d)
        hist = hist[-(max_order - 1):]
    return x


def ipndm_v_coeffs(t, i, order):
    """k_diffusion/sampling.py:891-921: variable-step Adams-Bashforth weights for nodes t[i], t[i-1], ..."""
    h_n = t[i + 1] - t[i]
    if order == 1:
        return [1.0]
    h_1 = t[i] - t[i - 1]
    if order == 2:
        return [(2 + h_n / h_1) / 2, -(h_n / h_1) / 2]
    h_2 = t[i - 1] - t[i - 2]
    temp1 = (1 - h_n / (3 * (h_n + h_1)) * (h_n * (h_n + h_1)) / (h_1 * (h_1 + h_2))) / 2
    if order == 3:
        return [(2 + h_n / h_1) / 2 + temp1, -(h_n / h_1) / 2 - (1 + h_1 / h_2) * temp1, temp1 * h_1 / h_2]
    h_3 = t[i - 2] - t[i - 3]
    temp2 = ((1 - h_n / (3 * (h_n + h_1))) / 2 + (1 - h_n / (2 * (h_n + h_1))) * h_n / (6 * (h_n + h_1 + h_2))) \
        * (h_n * (h_n + h_1) * (h_n + h_1 + h_2)) / (h_1 * (h_1 + h_2) * (h_1 + h_2 + h_3))
    q = h_1 * (h_1 + h_2) / (h_2 * (h_2 + h_3))
    return [(2 + h_n / h_1) / 2 + temp1 + temp2,
            -(h_n / h_1) / 2 - (1 + h_1 / h_2) * temp1 - (1 + h_1 / h_2 + q) * temp2,
            temp1 * h_1 / h_2 + (h_1 / h_2 + q * (1 + h_2 / h_3)) * temp2,
            -temp2 * q * h_1 / h_2]


def sample_ipndm_v(model, x, sigmas, max_order=4, callback=None):
    """k_diffusion/sampling.py:869-929."""
    s_in = x.new_ones([x.shape[0]])
    hist = []
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        d = (x - denoised) / sigmas[i]
        order = min(max_order, i + 1)
        c = ipndm_v_coeffs(sigmas, i, order)
        comb = c[0] * d
        for k in range(1, order):
            comb = comb + c[k] * hist[-k]
        x = x + (sigmas[i + 1] - sigmas[i]) * comb
        hist.append(d)
        hist = hist[-(max_order - 1):]
    return x


def deis_coeff_list(sigmas, max_order, n=10000):
    """k_diffusion/deis.py:13-21,59-88 ('tab' mode): sigma -> VP time t (beta_d, beta_min fitted to sigma_min 0.002, sigma_max 80,
    eps_s 1e-3), then per step the integral over [t_i, t_i+1] of  -1/2 dlog(alpha)/dtau / sqrt(alpha (1 - alpha))  times each Lagrange
    basis on the last `order` nodes, by an n-point Riemann sum.  dlog(alpha)/dtau = -tau (b1 - b0) - b0 in closed form (the
    reference differentiates it with autograd, deis.py:43-56)."""
    import numpy as np
    eps_s, s_min, s_max = 1e-3, torch.tensor(0.002), torch.tensor(80.0)
    beta_d = 2 * (np.log(s_min ** 2 + 1) / eps_s - np.log(s_max ** 2 + 1)) / (eps_s - 1)
    beta_min = np.log(s_max ** 2 + 1) - 0.5 * beta_d
    t = ((beta_min ** 2 + 2 * beta_d * (sigmas.clone().cpu() ** 2 + 1).log()).sqrt() - beta_min) / beta_d
    b0, b1 = beta_min, beta_d + beta_min
    out = []
    for i in range(len(t) - 1):
        order = min(i + 1, max_order)
        if order == 1:
            out.append([])
            continue
        taus = torch.linspace(t[i], t[i + 1], n)
        dtau = (t[i + 1] - t[i]) / n
        prev = t[[i - k for k in range(order)]]
        alpha = torch.exp(-0.5 * taus ** 2 * (b1 - b0) - taus * b0)
        integrand = -0.5 * (-taus * (b1 - b0) - b0) / torch.sqrt(alpha * (1 - alpha))
        cs = []
        for j in range(order):
            poly = 1
            for k in range(order):
                if k != j:
                    poly = poly * (taus - prev[k]) / (prev[j] - prev[k])
            cs.append(torch.sum(integrand * poly) * dtau)
        out.append(cs)
    return out


def sample_deis(model, x, sigmas, max_order=3, callback=None):
    """k_diffusion/sampling.py:933-981."""
    s_in = x.new_ones([x.shape[0]])
    coeffs = deis_coeff_list(sigmas, max_order)
    hist = []
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        d = (x - denoised) / sigmas[i]
        order = 1 if sigmas[i + 1] <= 0 else min(max_order, i + 1)
        if order == 1:
            x = x + (sigmas[i + 1] - sigmas[i]) * d
        else:
            c = coeffs[i]
            upd = c[0] * d
            for k in range(1, order):
                upd = upd + c[k] * hist[-k]
            x = x + upd
        hist.append(d)
        hist = hist[-(max_order - 1):]
    return x


def restart_step_list(sigmas):
    """modules/sd_samplers_extra.py:39-67: (possibly re-made Karras) schedule + the (old, new) sigma pairs incl. restart segments."""
    steps = sigmas.shape[0] - 1
    restart_list = {}
    if steps >= 20:
        restart_steps, restart_times = 9, 1
        if steps >= 36:
            restart_steps, restart_times = steps // 4, 2
        sigmas = get_sigmas_karras(steps - restart_steps * restart_times, sigmas[-2].item(), sigmas[0].item())
        restart_list = {0.1: [restart_steps + 1, restart_times, 2]}
    restart_list = {int(torch.argmin(abs(sigmas - k), dim=0)): v for k, v in restart_list.items()}
    pairs = []
    for i in range(len(sigmas) - 1):
        pairs.append((sigmas[i], sigmas[i + 1]))
        if i + 1 in restart_list:
            r_steps, r_times, r_max = restart_list[i + 1]
            min_idx, max_idx = i + 1, int(torch.argmin(abs(sigmas - r_max), dim=0))
            if max_idx < min_idx:
                seg = get_sigmas_karras(r_steps, sigmas[min_idx].item(), sigmas[max_idx].item())[:-1]
                for _ in range(r_times):
                    pairs.extend(zip(seg[:-1], seg[1:]))
    return pairs


def sample_restart(model, x, sigmas, noise_fn, s_noise=1.0, callback=None):
    """modules/sd_samplers_extra.py:7-74: Heun steps over restart_step_list, re-noising when the next pair starts higher."""
    s_in = x.new_ones([x.shape[0]])
    last = None
    for step_id, (old, new) in enumerate(restart_step_list(sigmas)):
        if last is not None and last < old:
            x = x + noise_fn() * s_noise * (old ** 2 - last ** 2) ** 0.5
        denoised = model(x, old * s_in)
        d = _to_d(x, old, denoised)
        _cb(callback, x, step_id, new, old, denoised)
        dt = new - old
        if new == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            d_2 = _to_d(x_2, new, model(x_2, new * s_in))
            x = x + (d + d_2) / 2 * dt
        last = new
    return x


# name -> (function, default scheduler, draws randn_like each step, ancestral noise, discard_next_to_last_sigma)
SAMPLERS_EXTRA = {
    "Heun": (sample_heun, None, True, False, False), "DPM2": (sample_dpm_2, "karras", True, False, True),
    "DPM2 a": (sample_dpm_2_ancestral, "karras", False, True, True), "DPM++ 2S a": (sample_dpmpp_2s_ancestral, "karras", False, True, False),
    "LMS": (sample_lms, None, False, False, False), "HeunPP2": (sample_heunpp2, None, True, False, False),
    "IPNDM": (sample_ipndm, None, False, False, False), "IPNDM_V": (sample_ipndm_v, None, False, False, False),
    "DEIS": (sample_deis, None, False, False, False), "Restart": (sample_restart, "karras", False, True, False),
}


# ---- timestep-space samplers (modules/sd_samplers_timesteps_impl.py) and the classic_ddim_eps_estimation wrapper ----------------------
def timesteps_for(steps):
    return torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)  # sd_samplers_timesteps.py:62


class EpsFromDenoiser:
    """modules/sd_samplers_cfg_denoiser.py:163-169, 201-202, 224-226: model(x_vp, t) -> eps via the sigma-space denoiser.
    `denoiser(x, sigma_vec)` returns denoised, or (denoised, cond_pred, uncond_pred)."""

    def __init__(self, denoiser, alphas_cumprod):
        self.denoiser, self.acd = denoiser, alphas_cumprod
        self.need_last_noise_uncond, self.last_noise_uncond = False, None

    def __call__(self, x, t):
        fake_sigmas = ((1 - self.acd) / self.acd) ** 0.5
        sigma = fake_sigmas[t.round().long().clip(0, int(fake_sigmas.shape[0]))]
        x = x * ((sigma ** 2.0 + 1.0) ** 0.5)[:, None, None, None]
        out = self.denoiser(x, sigma)
        denoised, uncond_pred = (out[0], out[2]) if isinstance(out, tuple) else (out, None)
        if self.need_last_noise_uncond:
            self.last_noise_uncond = (x - uncond_pred) / sigma[:, None, None, None]
        return (x - denoised) / sigma[:, None, None, None]


def _ddim_tables(acd, timesteps, eta):
    import numpy as np
    alphas = acd[timesteps]
    alphas_prev = acd[torch.nn.functional.pad(timesteps[:-1], pad=(1, 0))].to(torch.float64)
    sigmas = eta * np.sqrt((1 - alphas_prev.numpy()) / (1 - alphas) * (1 - alphas / alphas_prev.numpy()))
    return alphas, alphas_prev, torch.sqrt(1 - alphas), sigmas


def sample_ddim(eps_model, x, timesteps, noise_fn, eta=0.0, cfgpp=False):
    """sd_samplers_timesteps_impl.py:11-42 (ddim) / :45-83 (ddim_cfgpp: direction from the unconditional eps)."""
    alphas, alphas_prev, s1m, sigmas = _ddim_tables(eps_model.acd, timesteps, eta)
    eps_model.need_last_noise_uncond = cfgpp
    s_in, s_x = x.new_ones((x.shape[0])), x.new_ones((x.shape[0], 1, 1, 1))
    for i in range(len(timesteps) - 1):
        index = len(timesteps) - 1 - i
        e_t = eps_model(x, timesteps[index].item() * s_in)
        a_t, a_prev = alphas[index].item() * s_x, alphas_prev[index].item() * s_x
        sigma_t, sqrt_one_minus_at = sigmas[index].item() * s_x, s1m[index].item() * s_x
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * (eps_model.last_noise_uncond if cfgpp else e_t)
        x = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise_fn()
    return x


def sample_plms(eps_model, x, timesteps):
    """sd_samplers_timesteps_impl.py:86-142."""
    alphas, alphas_prev, s1m, _ = _ddim_tables(eps_model.acd, timesteps, 0.0)
    s_in, s_x = x.new_ones([x.shape[0]]), x.new_ones((x.shape[0], 1, 1, 1))
    old = []

    def x_prev_of(e, index):
        a_t, a_prev = alphas[index].item() * s_x, alphas_prev[index].item() * s_x
        pred_x0 = (x - s1m[index].item() * s_x * e) / a_t.sqrt()
        return a_prev.sqrt() * pred_x0 + (1.0 - a_prev).sqrt() * e

    for i in range(len(timesteps) - 1):
        index = len(timesteps) - 1 - i
        e_t = eps_model(x, timesteps[index].item() * s_in)
        if len(old) == 0:
            e_next = eps_model(x_prev_of(e_t, index), timesteps[max(index - 1, 0)].item() * s_in)
            e_prime = (e_t + e_next) / 2
        elif len(old) == 1:
            e_prime = (3 * e_t - old[-1]) / 2
        elif len(old) == 2:
            e_prime = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        x_new = x_prev_of(e_prime, index)
        old = (old + [e_t])[-3:]
        x = x_new
    return x


# ---- LCM (modules/sd_samplers_lcm.py) and DDPM (backend/modules/k_diffusion_extra.py) -------------------------------------------------
class LcmSchedule:
    """sd_samplers_lcm.py:10-49 + k_diffusion/external.py:76-120: the 50 training timesteps of LCM (every 20th of 1000)."""

    def __init__(self, predictor, timesteps=1000, original_timesteps=50):
        self.skip = timesteps // original_timesteps
        acd = 1.0 / (predictor.sigmas ** 2.0 + 1.0)
        valid = torch.zeros(original_timesteps)
        for x in range(original_timesteps):
            valid[original_timesteps - 1 - x] = acd[timesteps - 1 - x * self.skip]
        self.sigmas = ((1 - valid) / valid) ** 0.5
        self.log_sigmas = self.sigmas.log()

    def sigma_to_t(self, sigma):
        d = sigma.log() - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape) * self.skip + (self.skip - 1)

    def t_to_sigma(self, timestep):
        t = torch.clamp(((timestep - (self.skip - 1)) / self.skip).float(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()

    def get_sigmas(self, n):
        start, end = self.sigma_to_t(self.sigmas[-1]), self.sigma_to_t(self.sigmas[0])
        return append_zero(self.t_to_sigma(torch.linspace(start, end, n)))


def sample_lcm(model, x, sigmas, noise_fn, callback=None):
    """sd_samplers_lcm.py:69-83."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        x = denoised
        if sigmas[i + 1] > 0:
            x = x + sigmas[i + 1] * noise_fn()
    return x


def sample_ddpm(model, x, sigmas, noise_fn, callback=None):
    """backend/modules/k_diffusion_extra.py:12-42."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        _cb(callback, x, i, sigmas[i], sigmas[i], denoised)
        sigma, sigma_prev = sigmas[i], sigmas[i + 1]
        xv, noise = x / torch.sqrt(1.0 + sigma ** 2.0), (x - denoised) / sigma
        alpha_cumprod, alpha_cumprod_prev = 1 / (sigma * sigma + 1), 1 / (sigma_prev * sigma_prev + 1)
        alpha = alpha_cumprod / alpha_cumprod_prev
        mu = (1.0 / alpha).sqrt() * (xv - (1 - alpha) * noise / (1 - alpha_cumprod).sqrt())
        if sigma_prev > 0:
            mu = mu + ((1 - alpha) * (1.0 - alpha_cumprod_prev) / (1.0 - alpha_cumprod)).sqrt() * noise_fn()
        x = mu * torch.sqrt(1.0 + sigma_prev ** 2.0) if sigma_prev != 0 else mu
    return x


# ---- SDE family (noise_sampler injected: the reference's default is torchsde's BrownianTree) and DPM-Solver fast / adaptive ----------
def sample_dpmpp_sde(model, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0, r=0.5):
    """k_diffusion/sampling.py:607-645."""
    s_in = x.new_ones([x.shape[0]])
    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda s: s.log().neg()
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if sigmas[i + 1] == 0:
            x = x + _to_d(x, sigmas[i], denoised) * (sigmas[i + 1] - sigmas[i])
            continue
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(sd)
        x_2 = (sigma_fn(s_) / sigma_fn(t)) * x - (t - s_).expm1() * denoised
        x_2 = x_2 + noise_sampler(sigma_fn(t), sigma_fn(s)) * s_noise * su
        denoised_2 = model(x_2, sigma_fn(s) * s_in)
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(sd)
        denoised_d = (1 - fac) * denoised + fac * denoised_2
        x = (sigma_fn(t_next_) / sigma_fn(t)) * x - (t - t_next_).expm1() * denoised_d
        x = x + noise_sampler(sigma_fn(t), sigma_fn(t_next)) * s_noise * su
    return x


def sample_dpmpp_2m_sde(model, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0, solver_type="midpoint"):
    """k_diffusion/sampling.py:675-717."""
    s_in = x.new_ones([x.shape[0]])
    old, h_last = None, None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            eta_h = eta * h
            x = sigmas[i + 1] / sigmas[i] * (-eta_h).exp() * x + (-h - eta_h).expm1().neg() * denoised
            if old is not None:
                r = h_last / h
                if solver_type == "heun":
                    x = x + ((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r) * (denoised - old)
                else:
                    x = x + 0.5 * (-h - eta_h).expm1().neg() * (1 / r) * (denoised - old)
            if eta:
                x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise
            h_last = h
        old = denoised
    return x


def sample_dpmpp_3m_sde(model, x, sigmas, noise_sampler, eta=1.0, s_noise=1.0):
    """k_diffusion/sampling.py:721-768."""
    s_in = x.new_ones([x.shape[0]])
    den_1, den_2, h_1, h_2 = None, None, None, None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if sigmas[i + 1] == 0:
            x = denoised
        else:
            t, s = -sigmas[i].log(), -sigmas[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            x = torch.exp(-h_eta) * x + (-h_eta).expm1().neg() * denoised
            if h_2 is not None:
                r0, r1 = h_1 / h, h_2 / h
                d1_0, d1_1 = (denoised - den_1) / r0, (den_1 - den_2) / r1
                d1 = d1_0 + (d1_0 - d1_1) * r0 / (r0 + r1)
                d2 = (d1_0 - d1_1) / (r0 + r1)
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                x = x + phi_2 * d1 - phi_3 * d2
            elif h_1 is not None:
                x = x + (h_eta.neg().expm1() / h_eta + 1) * ((denoised - den_1) / (h_1 / h))
            if eta:
                x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * sigmas[i + 1] * (-2 * h * eta).expm1().neg().sqrt() * s_noise
            h_1, h_2 = h, h_1
        den_1, den_2 = denoised, den_1
    return x


class DpmSolver:
    """k_diffusion/sampling.py:397-463 (the eps cache is made explicit: a stage eps is passed in when an earlier solver step already has it)."""

    def __init__(self, model):
        self.model = model

    @staticmethod
    def sigma(t):
        return t.neg().exp()

    def eps(self, x, t):
        return (x - self.model(x, self.sigma(t) * x.new_ones([x.shape[0]]))) / self.sigma(t)

    def step1(self, x, t, t_next, eps):
        return x - self.sigma(t_next) * (t_next - t).expm1() * eps

    def step2(self, x, t, t_next, eps, r1=0.5):
        h = t_next - t
        s1 = t + r1 * h
        e1 = self.eps(x - self.sigma(s1) * (r1 * h).expm1() * eps, s1)
        return x - self.sigma(t_next) * h.expm1() * eps - self.sigma(t_next) / (2 * r1) * h.expm1() * (e1 - eps), e1

    def step3(self, x, t, t_next, eps, e1=None, r1=1 / 3, r2=2 / 3):
        h = t_next - t
        s1, s2 = t + r1 * h, t + r2 * h
        if e1 is None:
            e1 = self.eps(x - self.sigma(s1) * (r1 * h).expm1() * eps, s1)
        u2 = x - self.sigma(s2) * (r2 * h).expm1() * eps - self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (e1 - eps)
        e2 = self.eps(u2, s2)
        return x - self.sigma(t_next) * h.expm1() * eps - self.sigma(t_next) / r2 * (h.expm1() / h - 1) * (e2 - eps)

    def target(self, t, t_next, t_end, eta):
        if not eta:
            return t_next, 0.0
        sd, su = get_ancestral_step(self.sigma(t), self.sigma(t_next), eta)
        t_ = torch.minimum(t_end, -sd.log())
        return t_, (self.sigma(t_next) ** 2 - self.sigma(t_) ** 2) ** 0.5


def sample_dpm_fast(model, x, sigma_min, sigma_max, n, noise_fn, eta=0.0, s_noise=1.0):
    """k_diffusion/sampling.py:465-499, 546-555."""
    import math
    sol = DpmSolver(model)
    t_start, t_end = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min).log()
    m = math.floor(n / 3) + 1
    ts = torch.linspace(t_start, t_end, m + 1)
    orders = [3] * (m - 2) + [2, 1] if n % 3 == 0 else [3] * (m - 1) + [n % 3]
    for i, order in enumerate(orders):
        t, t_next = ts[i], ts[i + 1]
        t_, su = sol.target(t, t_next, t_end, eta)
        eps = sol.eps(x, t)
        x = sol.step1(x, t, t_, eps) if order == 1 else sol.step2(x, t, t_, eps)[0] if order == 2 else sol.step3(x, t, t_, eps)
        x = x + su * s_noise * noise_fn()
    return x


def sample_dpm_adaptive(model, x, sigma_min, sigma_max, noise_fn, order=3, rtol=0.05, atol=0.0078, h_init=0.05, eta=0.0, s_noise=1.0,
                        accept_safety=0.81):
    """k_diffusion/sampling.py:501-543, 558-569 with the PID controller (:368-394) at pcoeff 0, icoeff 1, dcoeff 0."""
    import math
    sol = DpmSolver(model)
    t_start, t_end = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min).log()
    s, x_prev, h = t_start, x, abs(h_init)
    b1 = 1.0 / (1.5 if eta else order)
    info = {"steps": 0, "n_accept": 0, "n_reject": 0}
    while s < t_end - 1e-5:
        t = torch.minimum(t_end, s + h)
        t_, su = sol.target(s, t, t_end, eta)
        eps = sol.eps(x, s)
        if order == 2:
            x_low = sol.step1(x, s, t_, eps)
            x_high, _ = sol.step2(x, s, t_, eps)
        else:
            x_low, e1 = sol.step2(x, s, t_, eps, r1=1 / 3)
            x_high = sol.step3(x, s, t_, eps, e1=e1)  # dpm_solver_3_step reuses the cached eps_r1 of the r1 = 1/3 two-stage step (:527-528)
        delta = torch.maximum(torch.tensor(atol), torch.tensor(rtol) * torch.maximum(x_low.abs(), x_prev.abs()))
        error = torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5
        factor = 1 + math.atan((1 / (float(error) + 1e-8)) ** b1 - 1)
        if factor >= accept_safety:
            x_prev = x_low
            x = x_high + su * s_noise * noise_fn()
            s = t
            info["n_accept"] += 1
        else:
            info["n_reject"] += 1
        h *= factor
        info["steps"] += 1
    return x, info
