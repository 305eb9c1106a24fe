"""Sigma schedule / prediction algebra of the path -- host-side mirror of backend/modules/k_prediction.py.

`Prediction` keeps the reference's names and semantics (:113-159): 1000-entry scaled-linear sigma table computed
in float64 and stored fp32, nearest-index `timestep()`, log-linear `sigma()`.  The table lives on the HOST: it is
1000 floats consulted once per step, while the tensor-sized arithmetic (`calculate_input`, `calculate_denoised`,
`noise_scaling`, :74-104) runs in the fused HIP kernels (fmx_unet_pack_input / fmx_cfg_combine / fmx_scale_f32).
"""
import torch

from ... import hipops as ops


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """k_prediction.py:18-39, float64."""
    if schedule == "linear":
        return torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    if schedule == "cosine":
        import math
        ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        return torch.clamp(1 - alphas[1:] / alphas[:-1], min=0, max=0.999)
    if schedule == "sqrt_linear":
        return torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    if schedule == "sqrt":
        return torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def rescale_zero_terminal_snr_sigmas(sigmas):
    """k_prediction.py:48-63: shift / rescale sqrt(alpha_bar) so that the last timestep has zero SNR (then clamp it to a finite sigma)."""
    alphas_bar_sqrt = (1 / ((sigmas * sigmas) + 1)).sqrt()
    first, last = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - last) * (first / (first - last))
    alphas_bar = alphas_bar_sqrt ** 2
    alphas_bar[-1] = 4.8973451890853435e-08
    return ((1 - alphas_bar) / alphas_bar) ** 0.5


class AbstractPrediction:
    def __init__(self, sigma_data=1.0, prediction_type="epsilon"):
        assert prediction_type in ("epsilon", "const", "v_prediction", "edm")  # k_prediction.py:71
        self.sigma_data = sigma_data
        self.prediction_type = prediction_type

    def noise_scaling(self, sigma, noise, latent_image=None, max_denoise=False):
        """k_prediction.py:94-104; `sigma` is a host scalar (sigmas[0]).  latent_image None == zeros (txt2img)."""
        s = float(sigma)
        if self.prediction_type == "const":  # :95-96  sigma * noise + (1 - sigma) * latent
            out = ops.scale_f32(noise, s)
            if latent_image is not None:
                out += (1.0 - s) * latent_image
            return out
        f = (1.0 + s ** 2.0) ** 0.5 if max_denoise else s
        out = ops.scale_f32(noise, f)
        if latent_image is not None:
            out += latent_image
        return out


class Prediction(AbstractPrediction):
    def __init__(self, sigma_data=1.0, prediction_type="epsilon", beta_schedule="linear", linear_start=0.00085,
                 linear_end=0.012, timesteps=1000):
        super().__init__(sigma_data, prediction_type)
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end)
        alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
        self.alphas_cumprod = alphas_cumprod.float()
        self.set_sigmas(sigmas)

    def set_sigmas(self, sigmas):
        """(also the hook a zero-terminal-SNR checkpoint uses: set_sigmas(rescale_zero_terminal_snr_sigmas(self.sigmas)))"""
        self.sigmas = sigmas.float()
        self.log_sigmas = sigmas.log().float()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        """Nearest table index (argmin in log space), :148-151.  `sigma`: host tensor [B] (or python floats)."""
        sigma = torch.as_tensor(sigma, dtype=torch.float32).cpu()
        dists = sigma.log().reshape(1, -1) - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)  # keeps a 0-dim input 0-dim (sd_schedulers.py:33-34 feeds it to linspace)

    def sigma(self, timestep):
        t = torch.clamp(torch.as_tensor(timestep).float().cpu(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()

    def percent_to_sigma(self, percent):
        if percent <= 0.0:
            return 999999999.9
        if percent >= 1.0:
            return 0.0
        return self.sigma(torch.tensor((1.0 - percent) * 999.0)).item()  # :166-167


class PredictionFlux(AbstractPrediction):
    """k_prediction.py:280-324: flow-matching 'const' prediction; sigma table = time-shifted linspace(1/N..1), mu from the
    image sequence length (diffusers calculate_shift, restated); timestep(sigma) = sigma."""

    def __init__(self, seq_len=4096, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15, pseudo_timestep_range=10000, mu=None):
        super().__init__(sigma_data=1.0, prediction_type="const")
        import math
        if mu is None:
            m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
            mu = seq_len * m + (base_shift - m * base_seq_len)
        self.mu = mu
        t = torch.arange(1, pseudo_timestep_range + 1, 1) / pseudo_timestep_range
        self.sigmas = (math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)).float()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        return sigma

    def sigma(self, timestep):
        return timestep

    def percent_to_sigma(self, percent):
        return 1.0 if percent <= 0.0 else 0.0 if percent >= 1.0 else 1.0 - percent
