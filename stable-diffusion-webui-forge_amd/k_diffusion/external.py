"""`ForgeScheduleLinker` -- mirror of k_diffusion/external.py:41-73 (sigma table <-> timestep via the predictor)."""
import torch

from . import sampling


class ForgeScheduleLinker:
    def __init__(self, predictor):
        self.predictor = predictor
        self.inner_model = None

    @property
    def sigmas(self):
        return self.predictor.sigmas

    @property
    def log_sigmas(self):
        return self.predictor.sigmas.log()

    @property
    def sigma_min(self):
        return self.predictor.sigma_min

    @property
    def sigma_max(self):
        return self.predictor.sigma_max

    def get_sigmas(self, n=None):
        if n is None:
            return sampling.append_zero(self.sigmas.flip(0))
        t_max = len(self.sigmas) - 1
        t = torch.linspace(t_max, 0, n)
        return sampling.append_zero(self.t_to_sigma(t))

    def sigma_to_t(self, sigma, quantize=None):
        return self.predictor.timestep(sigma)

    def t_to_sigma(self, t):
        return self.predictor.sigma(t)
