"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates backend/modules/k_prediction.py:74-104 (calculate_input / calculate_denoised / noise_scaling,
epsilon prediction), :113-159 (scaled-linear beta schedule -> 1000-entry sigma table, nearest-index
`timestep`, log-linear `sigma`) and backend/modules/k_model.py:25-46 (`apply_model`).
"""
import torch


class Predictor:
    def __init__(self, linear_start=0.00085, linear_end=0.012, timesteps=1000, sigma_data=1.0):
        # k_prediction.py:20-22: float64 linspace of sqrt(beta), squared; :127-133 cumprod -> sigmas, stored fp32
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
        self.sigmas = sig.float()
        self.log_sigmas = sig.log().float()
        self.sigma_data = sigma_data

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        # :148-151 nearest table entry in log space -> integer index
        d = sigma.log().reshape(1, -1) - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, t):
        # :153-159 log-linear interpolation
        t = torch.clamp(t.float(), 0, len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()

    def percent_to_sigma(self, percent):
        # k_prediction.py:161-167
        if percent <= 0.0:
            return 999999999.9
        if percent >= 1.0:
            return 0.0
        return self.sigma(torch.tensor(1000.0 * (1.0 - percent))).item()

    def calculate_input(self, sigma, x):
        s = sigma.view(-1, *([1] * (x.ndim - 1)))
        return x / (s ** 2 + self.sigma_data ** 2) ** 0.5

    def calculate_denoised(self, sigma, model_output, model_input):
        s = sigma.view(-1, *([1] * (model_output.ndim - 1)))
        return model_input - model_output * s

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        noise = noise * (torch.sqrt(1.0 + sigma ** 2.0) if max_denoise else sigma)
        return noise + latent_image


def apply_model(unet_fn, predictor, x, sigma, context, y=None):
    """k_model.py:25-46 with fp32 computation dtype: eps-net call wrapped in input/denoised algebra."""
    xc = predictor.calculate_input(sigma, x)
    t = predictor.timestep(sigma).float()
    eps = unet_fn(xc, t, context, y).float()
    return predictor.calculate_denoised(sigma, eps, x)
