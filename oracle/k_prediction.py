"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates backend/modules/k_prediction.py:74-104 (calculate_input / calculate_denoised / noise_scaling,
epsilon prediction), :113-159 (scaled-linear beta schedule -> 1000-entry sigma table, nearest-index
`timestep`, log-linear `sigma`) and backend/modules/k_model.py:25-46 (`apply_model`).
"""
import torch


def beta_schedule(schedule, n, linear_start, linear_end, cosine_s=8e-3):
    """k_prediction.py:18-39 (float64)."""
    import math
    if schedule == "linear":
        return torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2
    if schedule == "cosine":
        t = torch.arange(n + 1, dtype=torch.float64) / n + cosine_s
        a = torch.cos(t / (1 + cosine_s) * math.pi / 2).pow(2)
        a = a / a[0]
        return torch.clamp(1 - a[1:] / a[:-1], min=0, max=0.999)
    if schedule == "sqrt_linear":
        return torch.linspace(linear_start, linear_end, n, dtype=torch.float64)
    return torch.linspace(linear_start, linear_end, n, dtype=torch.float64) ** 0.5  # "sqrt"


def rescale_zero_terminal_snr_sigmas(sigmas):
    """k_prediction.py:48-63."""
    abs_ = (1 / ((sigmas * sigmas) + 1)).sqrt()
    a0, at = abs_[0].clone(), abs_[-1].clone()
    abs_ = (abs_ - at) * (a0 / (a0 - at))
    ab = abs_ ** 2
    ab[-1] = 4.8973451890853435e-08
    return ((1 - ab) / ab) ** 0.5


class Predictor:
    def __init__(self, linear_start=0.00085, linear_end=0.012, timesteps=1000, sigma_data=1.0, prediction_type="epsilon", schedule="linear"):
        # k_prediction.py:20-22: float64 linspace of sqrt(beta), squared; :127-133 cumprod -> sigmas, stored fp32
        betas = beta_schedule(schedule, timesteps, linear_start, linear_end)
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
        self.sigmas = sig.float()
        self.log_sigmas = sig.log().float()
        self.sigma_data = sigma_data
        self.prediction_type = prediction_type

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
        return self.sigma(torch.tensor((1.0 - percent) * 999.0)).item()

    def calculate_input(self, sigma, x):
        s = sigma.view(-1, *([1] * (x.ndim - 1)))
        return x / (s ** 2 + self.sigma_data ** 2) ** 0.5

    def calculate_denoised(self, sigma, model_output, model_input):
        s = sigma.view(-1, *([1] * (model_output.ndim - 1)))
        sd = self.sigma_data
        if self.prediction_type == "v_prediction":  # :83-86
            return model_input * sd ** 2 / (s ** 2 + sd ** 2) - model_output * s * sd / (s ** 2 + sd ** 2) ** 0.5
        if self.prediction_type == "edm":  # :87-90
            return model_input * sd ** 2 / (s ** 2 + sd ** 2) + model_output * s * sd / (s ** 2 + sd ** 2) ** 0.5
        return model_input - model_output * s

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        noise = noise * (torch.sqrt(1.0 + sigma ** 2.0) if max_denoise else sigma)
        return noise + latent_image


def apply_model(unet_fn, predictor, x, sigma, context, y=None, c_concat=None):
    """k_model.py:25-46 with fp32 computation dtype: eps-net call wrapped in input/denoised algebra.  c_concat (inpainting / edit
    models, :38-39) is concatenated AFTER the input scaling, i.e. unscaled."""
    xc = predictor.calculate_input(sigma, x)
    if c_concat is not None:
        xc = torch.cat([xc, c_concat], dim=1)
    t = predictor.timestep(sigma).float()
    eps = unet_fn(xc, t, context, y).float()
    return predictor.calculate_denoised(sigma, eps, x)
