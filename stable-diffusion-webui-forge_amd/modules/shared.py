"""The few globals of modules/shared.py the path reads (state flags, opts), without gradio.
Defaults follow modules/shared_options.py (randn_source :185, s_min_uncond :239, skip_early_cond :416,
eta_noise_seed_delta :408, always_discard_next_to_last_sigma :409, sgm_noise_multiplier :410,
use_old_karras_scheduler_sigmas :256, beta_dist_alpha/beta :417-418, uni_pc_* :411-414) and modules/shared_state.py."""
from types import SimpleNamespace

opts = SimpleNamespace(
    randn_source="CPU",  # reference default is "GPU"; "CPU"/"NV" are the device-independent sources (modules/rng.py:6-33)
    eta_noise_seed_delta=0, always_discard_next_to_last_sigma=False, sgm_noise_multiplier=False,
    use_old_karras_scheduler_sigmas=False, s_min_uncond=0.0, s_min_uncond_all=False, skip_early_cond=0.0,
    eta_ancestral=1.0, eta_ddim=0.0, sigma_min=0.0, sigma_max=0.0, rho=0.0, s_churn=0.0, s_tmin=0.0, s_tmax=0.0, s_noise=1.0,
    uni_pc_variant="bh1", uni_pc_skip_type="time_uniform", uni_pc_order=3, uni_pc_lower_order_final=True, beta_dist_alpha=0.6, beta_dist_beta=0.6, forge_try_reproduce="None", sd_vae_decode_method="Full",
)


class State:
    def __init__(self):
        self.interrupted = False
        self.skipped = False
        self.sampling_step = 0
        self.sampling_steps = 0
        self.current_latent = None

    def interrupt(self):
        self.interrupted = True


state = State()
sd_model = None
device = None
