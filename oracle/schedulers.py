"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates modules/sd_schedulers.py (the sigma-schedule table, :211-228) and the k-diffusion schedule helpers
(k_diffusion/sampling.py:19-45).  `linker` is anything with .sigmas (ascending table), .sigma_to_t(sigma), .t_to_sigma(t),
.get_sigmas(n) -- oracle.k_prediction.Predictor wrapped by `Linker` below (k_diffusion/external.py:41-73).
Pinned against the reference functions in tests/golden/schedulers.pt (oracle/make_golden.py gen_schedulers).
"""
import math

import numpy as np
import torch

from .sampling import append_zero, get_sigmas_karras, get_sigmas_linker


class Linker:
    def __init__(self, predictor):
        self.predictor = predictor
        self.sigmas = predictor.sigmas

    def get_sigmas(self, n):
        return get_sigmas_linker(self.predictor, n)

    def sigma_to_t(self, sigma):
        return self.predictor.timestep(sigma)

    def t_to_sigma(self, t):
        return self.predictor.sigma(t)


def get_sigmas_exponential(n, sigma_min, sigma_max):
    return append_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp())  # sampling.py:28-31


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    ramp = torch.linspace(1, 0, n) ** rho  # sampling.py:34-38
    return append_zero(torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)))


def uniform(n, sigma_min, sigma_max, linker):
    return linker.get_sigmas(n)  # sd_schedulers.py:28-29


def sgm_uniform(n, sigma_min, sigma_max, linker):
    start, end = linker.sigma_to_t(torch.tensor(sigma_max)), linker.sigma_to_t(torch.tensor(sigma_min))  # :32-40
    return torch.FloatTensor([linker.t_to_sigma(ts) for ts in torch.linspace(start, end, n + 1)[:-1]] + [0.0])


def _loglinear_interp(t_steps, num_steps):
    xs = np.linspace(0, 1, len(t_steps))  # :45-56
    ys = np.log(t_steps[::-1])
    return np.exp(np.interp(np.linspace(0, 1, num_steps), xs, ys))[::-1].copy()


AYS = {
    "align_your_steps": ([14.615, 6.475, 3.861, 2.697, 1.886, 1.396, 0.963, 0.652, 0.399, 0.152, 0.029],
                         [14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.029]),   # :58-62 (sd1, sdxl)
    "align_your_steps_GITS": ([14.615, 4.617, 2.507, 1.236, 0.702, 0.402, 0.240, 0.156, 0.104, 0.094, 0.029],
                              [14.615, 4.734, 2.567, 1.529, 0.987, 0.652, 0.418, 0.268, 0.179, 0.127, 0.029]),  # :149-152
}
AYS["align_your_steps_11"] = AYS["align_your_steps"]  # :176-179 repeats the table of :58-62


def align_your_steps(n, name, is_sdxl):
    sig = list(AYS[name][1 if is_sdxl else 0])
    sig = np.append(_loglinear_interp(sig, n), [0.0]) if n != len(sig) else sig + [0.0]  # :64-67
    return torch.FloatTensor(sig)


def kl_optimal(n, sigma_min, sigma_max):
    a_min, a_max = torch.arctan(torch.tensor(sigma_min)), torch.arctan(torch.tensor(sigma_max))  # :72-77
    idx = torch.arange(n + 1)
    return torch.tan(idx / n * a_min + (1.0 - idx / n) * a_max)


def simple(n, sigma_min, sigma_max, linker):
    ss = len(linker.sigmas) / n  # :80-86
    return torch.FloatTensor([float(linker.sigmas[-(1 + int(x * ss))]) for x in range(n)] + [0.0])


def normal(n, sigma_min, sigma_max, linker):
    start, end = linker.sigma_to_t(torch.tensor(sigma_max)), linker.sigma_to_t(torch.tensor(sigma_min))  # :89-104 (sgm False)
    return torch.FloatTensor([linker.t_to_sigma(ts) for ts in torch.linspace(start, end, n)] + [0.0])


def ddim(n, sigma_min, sigma_max, linker):
    ss = max(len(linker.sigmas) // n, 1)  # :107-116
    sigs = [float(linker.sigmas[x]) for x in range(1, len(linker.sigmas), ss)]
    return torch.FloatTensor(sigs[::-1] + [0.0])


def beta(n, sigma_min, sigma_max, linker, alpha=0.6, beta_=0.6):
    from scipy import stats
    ts = [stats.beta.ppf(x, alpha, beta_) for x in 1 - np.linspace(0, 1, n)]  # :119-127
    return torch.FloatTensor([sigma_min + x * (sigma_max - sigma_min) for x in ts] + [0.0])


def turbo(n, sigma_min, sigma_max, linker):
    ts = torch.flip(torch.arange(1, n + 1) * float(1000.0 / n) - 1, (0,)).round().long().clip(0, 999)  # :130-135
    return append_zero(linker.predictor.sigma(ts))


def get_sigmas(name, n, linker, is_sdxl=False, rho=None):
    """what modules/sd_samplers_kdiffusion.py:81-134 computes for an explicit scheduler name (default options)."""
    smin, smax = linker.sigmas[0].item(), linker.sigmas[-1].item()
    if name == "karras":
        return get_sigmas_karras(n, smin, smax, 7.0 if rho is None else rho)
    if name == "exponential":
        return get_sigmas_exponential(n, smin, smax)
    if name == "polyexponential":
        return get_sigmas_polyexponential(n, smin, smax, 1.0 if rho is None else rho)
    if name == "kl_optimal":
        return kl_optimal(n, smin, smax)
    if name in AYS:
        return align_your_steps(n, name, is_sdxl)
    return {"uniform": uniform, "sgm_uniform": sgm_uniform, "simple": simple, "normal": normal, "ddim": ddim, "beta": beta,
            "turbo": turbo}[name](n, smin, smax, linker)


ALL = ["uniform", "karras", "exponential", "polyexponential", "sgm_uniform", "kl_optimal", "align_your_steps", "simple", "normal",
       "ddim", "beta", "turbo", "align_your_steps_GITS", "align_your_steps_11"]
