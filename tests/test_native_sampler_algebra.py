"""CPU: the HOST side of the native samplers (schedule algebra -> per-step coefficients, RNG consumption, stage ordering) checked
against the reference's sampler functions on a closed-form denoiser (tests/golden/samplers_toy.pt, tiny_sd15_samples_more.pt,
tiny_sd15_samples_unipc.pt).  The four latent-sized kernels the loops launch are replaced HERE, for this test only, by their
one-line torch definitions (the kernels themselves are checked on the GPU in tests/test_gpu_kernels.py; the product never takes
this route: without the HIP library hipops raises)."""
import pytest
import torch

from forge_amd import hipops
from forge_amd.backend.modules import k_diffusion_extra
from forge_amd.k_diffusion import sampling as kd
from forge_amd.modules import sd_samplers_extra, sd_samplers_lcm, sd_samplers_timesteps_impl as ts_impl, shared
from oracle.k_prediction import Predictor
from oracle.make_golden import toy_denoiser, toy_inputs

from conftest import load_golden


@pytest.fixture(autouse=True)
def torch_kernels(monkeypatch):
    def lincomb(srcs, coefs, out=None):
        r = sum(float(c) * s for c, s in zip(coefs, srcs))
        return r if out is None else out.copy_(r)

    def euler_step(x, den, sigma, sigma_next, noise=None, noise_scale=0.0, out=None):
        r = x + (x - den) / sigma * (sigma_next - sigma)
        return r if noise is None else r + noise * noise_scale

    def lincomb3(x, d0, d1, a, b, c, out=None):
        return a * x + b * d0 + (c * d1 if d1 is not None else 0)
    monkeypatch.setattr(hipops, "lincomb", lincomb)
    monkeypatch.setattr(hipops, "euler_step", euler_step)
    monkeypatch.setattr(hipops, "lincomb3", lincomb3)
    monkeypatch.setattr(hipops, "scale_f32", lambda x, s, out=None: x * s)

    def error_norm(x_low, x_high, x_prev, atol, rtol):
        delta = torch.maximum(torch.tensor(atol), torch.tensor(rtol) * torch.maximum(x_low.abs(), x_prev.abs()))
        return float(torch.linalg.norm((x_low - x_high) / delta) / x_low.numel() ** 0.5)
    monkeypatch.setattr(hipops, "error_norm", error_norm)


def max_rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


class Seq:
    """TorchHijack stand-in: randn_like returns the next tensor of a fixed list and counts the draws."""

    def __init__(self, noises):
        self.noises, self.i = noises, 0

    def __getattr__(self, item):
        if item == "randn_like":
            def f(x):
                self.i += 1
                return self.noises[self.i - 1]
            return f
        return getattr(torch, item)


NATIVE = {"Heun": kd.sample_heun, "DPM2": kd.sample_dpm_2, "DPM2 a": kd.sample_dpm_2_ancestral, "DPM++ 2S a": kd.sample_dpmpp_2s_ancestral,
          "LMS": kd.sample_lms, "HeunPP2": kd.sample_heunpp2, "IPNDM": kd.sample_ipndm, "IPNDM_V": kd.sample_ipndm_v, "DEIS": kd.sample_deis,
          "Restart": sd_samplers_extra.restart_sampler}


@pytest.mark.parametrize("name", list(NATIVE))
def test_k_diffusion_family_host_algebra(name, monkeypatch):
    g = load_golden("samplers_toy.pt")
    x0, noises = toy_inputs()
    for steps in (5, 12, 24, 40):
        ref = g[(name, steps)]
        h = Seq(noises)
        monkeypatch.setattr(kd, "torch", h)
        got = NATIVE[name](toy_denoiser, x0 * ref["sigmas"][0], ref["sigmas"], disable=True)
        monkeypatch.setattr(kd, "torch", torch)
        assert max_rel(got, ref["latent"]) < 2e-5, (name, steps, max_rel(got, ref["latent"]))
        assert h.i == ref["draws"], (name, steps)


class _EpsModel:
    """CFGDenoiser's classic_ddim_eps_estimation arithmetic around the toy denoiser (the real class is exercised on the GPU)."""

    def __init__(self, acd):
        self.inner_model = type("M", (), {})()
        self.inner_model.inner_model = type("M", (), {"alphas_cumprod": acd})()
        self.need_last_noise_uncond, self.last_noise_uncond = False, None

    def __call__(self, x, t, **kw):
        fake = ((1 - self.inner_model.inner_model.alphas_cumprod) / self.inner_model.inner_model.alphas_cumprod) ** 0.5
        sigma = fake[t.round().long().clip(0, 999)]
        xs = x * ((sigma ** 2 + 1) ** 0.5)[:, None, None, None]
        den = toy_denoiser(xs, sigma)
        if self.need_last_noise_uncond:
            self.last_noise_uncond = (xs - toy_denoiser(0.5 * xs, sigma)) / sigma[:, None, None, None]
        return (xs - den) / sigma[:, None, None, None]


@pytest.mark.parametrize("label", ["DDIM", "DDIM eta", "DDIM CFG++", "PLMS", "LCM", "DDPM"])
def test_timestep_lcm_ddpm_host_algebra(label, monkeypatch):
    g = load_golden("tiny_sd15_samples_more.pt")[label]
    x0, noises = toy_inputs()
    h = Seq(noises)
    monkeypatch.setattr(kd, "torch", h)
    monkeypatch.setattr(k_diffusion_extra, "torch", h)
    acd = 1.0 / (Predictor().sigmas ** 2.0 + 1.0)
    if label in ("LCM", "DDPM"):
        fn = sd_samplers_lcm.sample_lcm if label == "LCM" else k_diffusion_extra.sample_ddpm
        got = fn(toy_denoiser, x0 * g["sigmas"][0], g["sigmas"], disable=True)
    else:
        fn = {"DDIM": ts_impl.ddim, "DDIM eta": ts_impl.ddim, "DDIM CFG++": ts_impl.ddim_cfgpp, "PLMS": ts_impl.plms}[label]
        kw = {"eta": g["eta"]} if "eta" in g else {}
        got = fn(_EpsModel(acd), x0.clone(), g["timesteps"], disable=True, callback=lambda d: None, **kw)
    assert max_rel(got, g["toy"]) < 2e-5
    assert h.i == g["toy_draws"]


def test_unipc_host_algebra_all_variants(monkeypatch):
    g = load_golden("tiny_sd15_samples_unipc.pt")
    x0, _ = toy_inputs()
    acd = 1.0 / (Predictor().sigmas ** 2.0 + 1.0)
    saved = {k: getattr(shared.opts, k) for k in ("uni_pc_variant", "uni_pc_skip_type", "uni_pc_order", "uni_pc_lower_order_final")}
    try:
        for (variant, skip, order, lof, steps), want in g["toy"].items():
            shared.opts.uni_pc_variant, shared.opts.uni_pc_skip_type = variant, skip
            shared.opts.uni_pc_order, shared.opts.uni_pc_lower_order_final = order, lof
            ts = torch.clip(torch.asarray(list(range(0, 1000, 1000 // steps))) + 1, 0, 999)
            xin = x0[:1].clone() if variant == "vary_coeff" else x0.clone()
            got = ts_impl.unipc(_EpsModel(acd), xin, ts, extra_args={}, callback=lambda d: None)
            assert max_rel(got, want) < 1e-4, (variant, skip, order, lof, steps, max_rel(got, want))
        for k, v in saved.items():
            setattr(shared.opts, k, v)
        ts = torch.clip(torch.asarray(list(range(0, 1000, 1000 // 6))) + 1, 0, 999)
        got = ts_impl.unipc(_EpsModel(acd), x0.clone(), ts[:4], extra_args={}, callback=lambda d: None, is_img2img=True)
        assert max_rel(got, g["toy_img2img"]) < 1e-4
    finally:
        for k, v in saved.items():
            setattr(shared.opts, k, v)


SDE_NATIVE = {"DPM++ SDE": kd.sample_dpmpp_sde, "DPM++ SDE eta0.5": kd.sample_dpmpp_sde, "DPM++ 2M SDE": kd.sample_dpmpp_2m_sde,
              "DPM++ 2M SDE Heun": kd.sample_dpmpp_2m_sde, "DPM++ 2M SDE eta0": kd.sample_dpmpp_2m_sde, "DPM++ 3M SDE": kd.sample_dpmpp_3m_sde,
              "DPM++ 3M SDE eta0": kd.sample_dpmpp_3m_sde}


@pytest.mark.parametrize("label", list(SDE_NATIVE))
def test_sde_family_host_algebra(label):
    from oracle.make_golden import ListNoiseSampler
    g = load_golden("samplers_sde_dpm.pt")
    x0, noises = toy_inputs()
    for steps in (5, 12):
        ref = g[(label, steps)]
        ns = ListNoiseSampler(noises)
        got = SDE_NATIVE[label](toy_denoiser, x0 * ref["sigmas"][0], ref["sigmas"], noise_sampler=ns, disable=True, **ref["kw"])
        assert max_rel(got, ref["latent"]) < 2e-5, (label, steps, max_rel(got, ref["latent"]))
        assert len(ns.asked) == len(ref["asked"]) and all(abs(a[0] - b[0]) < 1e-4 * b[0] and abs(a[1] - b[1]) < 1e-4 * b[1]
                                                          for a, b in zip(ns.asked, ref["asked"]))


def test_dpm_solver_fast_and_adaptive_host_algebra(monkeypatch):
    g = load_golden("samplers_sde_dpm.pt")
    x0, noises = toy_inputs()
    p = Predictor()
    smin, smax = p.sigmas[0].item(), p.sigmas[-1].item()
    for n in (5, 6, 7, 12):
        for eta in (0.0, 0.6):
            h = Seq(noises)
            monkeypatch.setattr(kd, "torch", h)
            got = kd.sample_dpm_fast(toy_denoiser, x0 * smax, smin, smax, n, disable=True, eta=eta)
            monkeypatch.setattr(kd, "torch", torch)
            ref = g[("DPM fast", n, eta)]
            assert max_rel(got, ref["latent"]) < 2e-5 and h.i == ref["draws"], (n, eta)
    for order in (2, 3):
        for eta in (0.0, 0.6):
            h = Seq(noises)
            monkeypatch.setattr(kd, "torch", h)
            got, info = kd.sample_dpm_adaptive(toy_denoiser, x0 * smax, smin, smax, disable=True, order=order, eta=eta, return_info=True)
            monkeypatch.setattr(kd, "torch", torch)
            ref = g[("DPM adaptive", order, eta)]
            assert info == ref["info"], (order, eta, info, ref["info"])
            assert max_rel(got, ref["latent"]) < 5e-5 and h.i == ref["draws"], (order, eta)
