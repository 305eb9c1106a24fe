"""Sigma-schedule table -- mirror of modules/sd_schedulers.py (`Scheduler` :18-26, the functions :28-209, `schedulers` /
`schedulers_map` :211-230).  All of it is O(steps) host arithmetic on the fp32 sigma table; the result stays on the host,
where the native sampler loops read it as python floats."""
import dataclasses

import numpy as np
import torch

from . import shared
from ..k_diffusion import sampling as kd_sampling


@dataclasses.dataclass
class Scheduler:
    name: str
    label: str
    function: any
    default_rho: float = -1
    need_inner_model: bool = False
    aliases: list = None


def uniform(n, sigma_min, sigma_max, inner_model, device):
    return inner_model.get_sigmas(n).to(device)


def _timestep_range(inner_model, sigma_min, sigma_max):
    return inner_model.sigma_to_t(torch.tensor(sigma_max)), inner_model.sigma_to_t(torch.tensor(sigma_min))


def sgm_uniform(n, sigma_min, sigma_max, inner_model, device):
    start, end = _timestep_range(inner_model, sigma_min, sigma_max)
    sigs = [inner_model.t_to_sigma(ts) for ts in torch.linspace(start, end, n + 1)[:-1]] + [0.0]
    return torch.FloatTensor(sigs).to(device)


def normal_scheduler(n, sigma_min, sigma_max, inner_model, device, sgm=False, floor=False):
    start, end = _timestep_range(inner_model, sigma_min, sigma_max)
    timesteps = torch.linspace(start, end, n + 1)[:-1] if sgm else torch.linspace(start, end, n)
    return torch.FloatTensor([inner_model.t_to_sigma(ts) for ts in timesteps] + [0.0]).to(device)


def simple_scheduler(n, sigma_min, sigma_max, inner_model, device):
    ss = len(inner_model.sigmas) / n
    sigs = [float(inner_model.sigmas[-(1 + int(x * ss))]) for x in range(n)] + [0.0]
    return torch.FloatTensor(sigs).to(device)


def ddim_scheduler(n, sigma_min, sigma_max, inner_model, device):
    table = inner_model.sigmas
    ss = max(len(table) // n, 1)
    sigs = [float(table[x]) for x in range(1, len(table), ss)][::-1] + [0.0]
    return torch.FloatTensor(sigs).to(device)


def beta_scheduler(n, sigma_min, sigma_max, inner_model, device):
    from scipy import stats
    alpha, beta = shared.opts.beta_dist_alpha, shared.opts.beta_dist_beta
    ts = [stats.beta.ppf(x, alpha, beta) for x in 1 - np.linspace(0, 1, n)]
    return torch.FloatTensor([sigma_min + x * (sigma_max - sigma_min) for x in ts] + [0.0]).to(device)


def turbo_scheduler(n, sigma_min, sigma_max, inner_model, device):
    predictor = inner_model.inner_model.forge_objects.unet.model.predictor
    timesteps = torch.flip(torch.arange(1, n + 1) * float(1000.0 / n) - 1, (0,)).round().long().clip(0, 999)
    sigmas = predictor.sigma(timesteps)
    return torch.cat([sigmas, sigmas.new_zeros([1])]).to(device)


def kl_optimal(n, sigma_min, sigma_max, device):
    alpha_min, alpha_max = torch.arctan(torch.tensor(sigma_min)), torch.arctan(torch.tensor(sigma_max))
    frac = torch.arange(n + 1) / n
    return torch.tan(frac * alpha_min + (1.0 - frac) * alpha_max).to(device)


# published "Align Your Steps" tables (sd_schedulers.py:58-62, 149-152, 176-179, 199-202): (SD1.x, SDXL)
_AYS_TABLES = {
    "ays": ([14.615, 6.475, 3.861, 2.697, 1.886, 1.396, 0.963, 0.652, 0.399, 0.152, 0.029],
            [14.615, 6.315, 3.771, 2.181, 1.342, 0.862, 0.555, 0.380, 0.234, 0.113, 0.029]),
    "gits": ([14.615, 4.617, 2.507, 1.236, 0.702, 0.402, 0.240, 0.156, 0.104, 0.094, 0.029],
             [14.615, 4.734, 2.567, 1.529, 0.987, 0.652, 0.418, 0.268, 0.179, 0.127, 0.029]),
    "ays32": ([14.615, 11.23951352, 8.64363081, 6.64729424, 5.57250862, 4.71648546, 3.99196065, 3.5195609, 3.13490466, 2.79228788,
               2.48773628, 2.21663865, 1.97508351, 1.7793172, 1.61475335, 1.46540953, 1.314849, 1.16642497, 1.03475547, 0.91573744,
               0.80748169, 0.71202361, 0.621739, 0.53065202, 0.4529096, 0.37491455, 0.27461819, 0.2011529, 0.14105873, 0.06682881,
               0.03166121, 0.015],
              [14.615, 11.1491618, 8.50522127, 6.48827151, 5.43707402, 4.60398619, 3.89854704, 3.27407457, 2.74396527, 2.29968659,
               1.95448514, 1.67108715, 1.42878152, 1.23181009, 1.06789649, 0.92579443, 0.80290886, 0.69660121, 0.60436903, 0.52852552,
               0.46773344, 0.41393379, 0.36258186, 0.31008517, 0.26518925, 0.22326461, 0.17653877, 0.13959192, 0.10587381, 0.05519369,
               0.02877334, 0.015]),
}


def _ays(table):
    def fn(n, sigma_min, sigma_max, device="cpu"):
        sigmas = list(_AYS_TABLES[table][1 if getattr(shared.sd_model, "is_sdxl", False) else 0])
        if n != len(sigmas):  # log-linear interpolation of the decreasing table to n points
            xs = np.linspace(0, 1, len(sigmas))
            ys = np.log(sigmas[::-1])
            sigmas = np.append(np.exp(np.interp(np.linspace(0, 1, n), xs, ys))[::-1].copy(), [0.0])
        else:
            sigmas.append(0.0)
        return torch.FloatTensor(sigmas).to(device)
    return fn


schedulers = [
    Scheduler("automatic", "Automatic", None),
    Scheduler("uniform", "Uniform", uniform, need_inner_model=True),
    Scheduler("karras", "Karras", kd_sampling.get_sigmas_karras, default_rho=7.0),
    Scheduler("exponential", "Exponential", kd_sampling.get_sigmas_exponential),
    Scheduler("polyexponential", "Polyexponential", kd_sampling.get_sigmas_polyexponential, default_rho=1.0),
    Scheduler("sgm_uniform", "SGM Uniform", sgm_uniform, need_inner_model=True, aliases=["SGMUniform"]),
    Scheduler("kl_optimal", "KL Optimal", kl_optimal),
    Scheduler("align_your_steps", "Align Your Steps", _ays("ays")),
    Scheduler("simple", "Simple", simple_scheduler, need_inner_model=True),
    Scheduler("normal", "Normal", normal_scheduler, need_inner_model=True),
    Scheduler("ddim", "DDIM", ddim_scheduler, need_inner_model=True),
    Scheduler("beta", "Beta", beta_scheduler, need_inner_model=True),
    Scheduler("turbo", "Turbo", turbo_scheduler, need_inner_model=True),
    Scheduler("align_your_steps_GITS", "Align Your Steps GITS", _ays("gits")),
    Scheduler("align_your_steps_11", "Align Your Steps 11", _ays("ays")),
    Scheduler("align_your_steps_32", "Align Your Steps 32", _ays("ays32")),
]

schedulers_map = {**{x.name: x for x in schedulers}, **{x.label: x for x in schedulers}}
