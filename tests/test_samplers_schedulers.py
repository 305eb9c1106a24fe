"""CPU: the rest of the sampler / scheduler tables.
  * oracle/sampling.py restatements vs the REFERENCE's sampler functions on a closed-form denoiser (tests/golden/samplers_toy.pt)
    and through the reference UNet + sampling_function (tests/golden/tiny_sd15_samples_extra.pt);
  * oracle/schedulers.py AND the product's host-side modules/sd_schedulers.py, k_diffusion/deis.py vs the reference's
    (tests/golden/schedulers.pt, samplers_toy.pt) -- bit-exact: it is the same fp32 / numpy host arithmetic."""
import pytest
import torch

from forge_amd import synth
from oracle import pipeline, sampling as osamp, schedulers as osched
from oracle.k_prediction import Predictor
from oracle.make_golden import toy_denoiser, toy_inputs

from conftest import load_golden

EXTRA = list(osamp.SAMPLERS_EXTRA)


def max_rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("name", EXTRA)
def test_oracle_sampler_restatements_vs_reference_functions(name):
    g = load_golden("samplers_toy.pt")
    x0, noises = toy_inputs()
    fn, _, draws, ancestral, _ = osamp.SAMPLERS_EXTRA[name]
    for steps in (5, 12, 24, 40):
        ref = g[(name, steps)]
        it = iter(noises)
        kw = {"noise_fn": lambda: next(it)} if (draws or ancestral) else {}
        got = fn(toy_denoiser, x0 * ref["sigmas"][0], ref["sigmas"], **kw)
        assert max_rel(got, ref["latent"]) < 5e-6, (name, steps)
        used = 64 - len(list(it))
        assert used == ref["draws"], (name, steps, used, ref["draws"])  # the RNG stream advances exactly as in the reference


@pytest.mark.parametrize("name", EXTRA)
def test_oracle_pipeline_extra_samplers_vs_reference_stack(name):
    g = load_golden("tiny_sd15_samples_extra.pt")
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(len(g["seeds"]), cfg["context_dim"], None, seed=1234)
    lat, sigmas = pipeline.txt2img_latents(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, g[name]["steps"], sampler_name=name)
    torch.testing.assert_close(sigmas, g[name]["sigmas"], rtol=0, atol=0)
    assert max_rel(lat, g[name]["latent"]) < 5e-4


def test_oracle_schedulers_bit_exact():
    g = load_golden("schedulers.pt")
    linker = osched.Linker(Predictor())
    for name in osched.ALL:
        for sdxl in (False, True):
            for n in (1, 4, 11, 20, 31, 32):
                want = g[(name, n, sdxl)]
                got = osched.get_sigmas(name, n, linker, is_sdxl=sdxl).float()
                assert want.shape == got.shape and torch.equal(want, got), (name, n, sdxl)


def test_product_schedulers_bit_exact():
    """modules/sd_schedulers.py of the product (host arithmetic, no GPU involved) against the reference's table."""
    from types import SimpleNamespace
    from forge_amd.backend.modules.k_prediction import Prediction
    from forge_amd.k_diffusion.external import ForgeScheduleLinker
    from forge_amd.modules import sd_schedulers, shared
    g = load_golden("schedulers.pt")
    pred = Prediction(prediction_type="epsilon", beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
    linker = ForgeScheduleLinker(pred)
    linker.inner_model = SimpleNamespace(forge_objects=SimpleNamespace(unet=SimpleNamespace(model=SimpleNamespace(predictor=pred))))
    assert {s.name: s.label for s in sd_schedulers.schedulers} == g["labels"]
    saved = shared.sd_model
    try:
        for sdxl in (False, True):
            shared.sd_model = SimpleNamespace(is_sdxl=sdxl)
            for sch in sd_schedulers.schedulers:
                if sch.function is None:
                    continue
                for n in (1, 4, 11, 20, 31, 32):
                    kw = {"sigma_min": pred.sigmas[0].item(), "sigma_max": pred.sigmas[-1].item()}
                    if sch.need_inner_model:
                        kw["inner_model"] = linker
                    got = sch.function(n=n, **kw, device="cpu").float()
                    want = g[(sch.name, n, sdxl)]
                    assert want.shape == got.shape and torch.equal(want, got), (sch.name, n, sdxl)
    finally:
        shared.sd_model = saved
    assert torch.equal(sd_schedulers.schedulers_map["Karras"].function(n=10, sigma_min=0.03, sigma_max=14.0, rho=5.0, device="cpu"), g["rho"]["karras_5"])
    assert torch.equal(sd_schedulers.schedulers_map["polyexponential"].function(n=10, sigma_min=0.03, sigma_max=14.0, rho=2.0, device="cpu"),
                       g["rho"]["polyexponential_2"])


def test_deis_coefficient_tables():
    from forge_amd.k_diffusion import deis
    g = load_golden("samplers_toy.pt")
    sig = g["deis_sigmas"]
    for key, order, mode in (("deis_tab_3", 3, "tab"), ("deis_tab_4", 4, "tab"), ("deis_rhoab_3", 3, "rhoab")):
        got = deis.get_deis_coeff_list(sig, order, deis_mode=mode)
        assert len(got) == len(g[key])
        for i, (a, b) in enumerate(zip(got, g[key])):
            assert len(a) == len(b), (key, i)
            if i == len(got) - 1 and mode == "tab":
                continue  # the last interval ends at sigma = 0 where the VP integrand is singular; sample_deis never reads it
            for x, y in zip(a, b):
                assert abs(float(x) - y) <= 2e-5 * max(abs(y), 1e-3), (key, i, float(x), y)
    ours = osamp.deis_coeff_list(sig, 3)
    for a, b in zip(ours[:-1], g["deis_tab_3"][:-1]):
        for x, y in zip(a, b):
            assert abs(float(x) - y) <= 2e-5 * max(abs(y), 1e-3)


def test_sampler_table_matches_reference_names_and_options():
    from forge_amd.modules import sd_samplers, sd_samplers_kdiffusion as kd
    names = {x.name: x for x in sd_samplers.all_samplers}
    for name in ["DPM++ 2M", "Euler a", "Euler"] + EXTRA:
        assert name in names, name
    assert names["DPM2"].options == {"scheduler": "karras", "discard_next_to_last_sigma": True, "second_order": True}
    assert names["DPM++ 2S a"].options["scheduler"] == "karras" and names["Restart"].options["scheduler"] == "karras"
    assert sd_samplers.find_sampler_config("k_dpm_2_a").name == "DPM2 a"
    assert set(kd.sampler_extra_params["sample_heun"]) == {"s_churn", "s_tmin", "s_tmax", "s_noise"}


MORE = [("DDIM", {}), ("DDIM eta", {"eta": 0.7}), ("DDIM CFG++", {}), ("PLMS", {}), ("LCM", {}), ("DDPM", {})]


def _toy_full(x, sigma):
    d = toy_denoiser(x, sigma)
    return d, d, toy_denoiser(0.5 * x, sigma)  # (denoised, cond_pred, stand-in uncond_pred), as oracle/make_golden.py ToyInner


@pytest.mark.parametrize("label,kw", MORE)
def test_oracle_timestep_lcm_ddpm_restatements_vs_reference_functions(label, kw):
    g = load_golden("tiny_sd15_samples_more.pt")[label]
    x0, noises = toy_inputs()
    it = iter(noises)
    nf = lambda: next(it)
    acd = 1.0 / (Predictor().sigmas ** 2.0 + 1.0)
    if label.startswith("DDIM"):
        got = osamp.sample_ddim(osamp.EpsFromDenoiser(_toy_full, acd), x0.clone(), g["timesteps"], nf, eta=kw.get("eta", 0.0), cfgpp=label == "DDIM CFG++")
    elif label == "PLMS":
        got = osamp.sample_plms(osamp.EpsFromDenoiser(_toy_full, acd), x0.clone(), g["timesteps"])
    else:
        fn = osamp.sample_lcm if label == "LCM" else osamp.sample_ddpm
        got = fn(toy_denoiser, x0 * g["sigmas"][0], g["sigmas"], nf)
    assert max_rel(got, g["toy"]) < 5e-6
    assert 64 - len(list(it)) == g["toy_draws"]


@pytest.mark.parametrize("label,kw", MORE)
def test_oracle_pipeline_more_samplers_vs_reference_stack(label, kw):
    g = load_golden("tiny_sd15_samples_more.pt")
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(len(g["seeds"]), cfg["context_dim"], None, seed=1234)
    name = "DDIM" if label == "DDIM eta" else label
    lat = pipeline.txt2img_latents_more(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, g[label]["steps"], name, eta=kw.get("eta", 0.0))
    assert max_rel(lat, g[label]["latent"]) < 5e-4


def test_lcm_schedule_bit_exact():
    """oracle AND product LCM schedules (50 training timesteps, sigma <-> t maps) vs the reference's DiscreteEpsDDPMDenoiser-based one."""
    from types import SimpleNamespace
    from forge_amd.backend.modules.k_prediction import Prediction
    from forge_amd.modules.sd_samplers_lcm import LCMCompVisDenoiser
    g = load_golden("tiny_sd15_samples_more.pt")
    pred = Prediction(prediction_type="epsilon", beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
    prod = LCMCompVisDenoiser(SimpleNamespace(forge_objects=SimpleNamespace(unet=SimpleNamespace(model=SimpleNamespace(predictor=pred)))))
    orc = osamp.LcmSchedule(Predictor())
    assert torch.equal(prod.sigmas, g["lcm_sigmas_table"]) and torch.equal(orc.sigmas, g["lcm_sigmas_table"])
    for n, want in g["lcm_get_sigmas"].items():
        assert torch.equal(prod.get_sigmas(n), want) and torch.equal(orc.get_sigmas(n), want), n


def test_ddpm_step_coefficients_match_the_written_out_step():
    from forge_amd.backend.modules.k_diffusion_extra import ddpm_step_coefficients
    x, den, nz = torch.randn(3, 4), torch.randn(3, 4), torch.randn(3, 4)
    for s, sp in ((14.6, 9.7), (1.0, 0.4), (0.1, 0.0)):
        sig = torch.tensor([s, sp])
        want = osamp.sample_ddpm(lambda xx, ss: den, x, sig, lambda: nz)
        cx, cd, cn = ddpm_step_coefficients(sig[0], sig[1])
        torch.testing.assert_close(cx * x + cd * den + cn * nz, want, rtol=1e-5, atol=1e-5)


def test_oracle_unipc_vs_reference_classes():
    """oracle/unipc.py vs the reference's unipc() / UniPC / NoiseScheduleVP: every variant x skip type x (order, lower_order_final) on the
    toy denoiser, the img2img start time, and the default configuration through the reference UNet stack."""
    from oracle import unipc as ou
    g = load_golden("tiny_sd15_samples_unipc.pt")
    x0, _ = toy_inputs()
    acd = 1.0 / (Predictor().sigmas ** 2.0 + 1.0)
    for (variant, skip, order, lof, steps), want in g["toy"].items():
        n = len(osamp.timesteps_for(steps))
        xin = x0[:1].clone() if variant == "vary_coeff" else x0.clone()
        got = ou.sample_unipc(osamp.EpsFromDenoiser(_toy_full, acd), xin, n, acd, variant, skip, order, lof)
        assert max_rel(got, want) < 3e-5, (variant, skip, order, lof, steps, max_rel(got, want))
    ts = osamp.timesteps_for(6)[:4]
    got = ou.sample_unipc(osamp.EpsFromDenoiser(_toy_full, acd), x0.clone(), len(ts), acd, t_start=ts[-1] / 1000 + 1 / 1000)
    assert max_rel(got, g["toy_img2img"]) < 3e-5
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(len(g["seeds"]), cfg["context_dim"], None, seed=1234)
    for steps in (6, 9):
        lat = pipeline.txt2img_latents_more(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, steps, "UniPC")
        assert max_rel(lat, g[steps]["latent"]) < 5e-4, steps


SDE_CASES = {"DPM++ SDE": osamp.sample_dpmpp_sde, "DPM++ SDE eta0.5": osamp.sample_dpmpp_sde, "DPM++ 2M SDE": osamp.sample_dpmpp_2m_sde,
             "DPM++ 2M SDE Heun": osamp.sample_dpmpp_2m_sde, "DPM++ 2M SDE eta0": osamp.sample_dpmpp_2m_sde,
             "DPM++ 3M SDE": osamp.sample_dpmpp_3m_sde, "DPM++ 3M SDE eta0": osamp.sample_dpmpp_3m_sde}


@pytest.mark.parametrize("label", list(SDE_CASES))
def test_oracle_sde_family_vs_reference_functions(label):
    from oracle.make_golden import ListNoiseSampler
    g = load_golden("samplers_sde_dpm.pt")
    x0, noises = toy_inputs()
    for steps in (5, 12):
        ref = g[(label, steps)]
        ns = ListNoiseSampler(noises)
        got = SDE_CASES[label](toy_denoiser, x0 * ref["sigmas"][0], ref["sigmas"], ns, **ref["kw"])
        assert max_rel(got, ref["latent"]) < 5e-6, (label, steps)
        assert len(ns.asked) == len(ref["asked"]) and all(abs(a[0] - b[0]) < 1e-4 * b[0] and abs(a[1] - b[1]) < 1e-4 * b[1]
                                                          for a, b in zip(ns.asked, ref["asked"])), (label, steps)


def test_oracle_dpm_solver_fast_and_adaptive_vs_reference_functions():
    g = load_golden("samplers_sde_dpm.pt")
    x0, noises = toy_inputs()
    p = Predictor()
    smin, smax = p.sigmas[0].item(), p.sigmas[-1].item()
    for n in (5, 6, 7, 12):
        for eta in (0.0, 0.6):
            it = iter(noises)
            got = osamp.sample_dpm_fast(toy_denoiser, x0 * smax, smin, smax, n, lambda: next(it), eta=eta)
            ref = g[("DPM fast", n, eta)]
            assert max_rel(got, ref["latent"]) < 5e-6, (n, eta)
            assert 64 - len(list(it)) == ref["draws"]
    for order in (2, 3):
        for eta in (0.0, 0.6):
            it = iter(noises)
            got, info = osamp.sample_dpm_adaptive(toy_denoiser, x0 * smax, smin, smax, lambda: next(it), order=order, eta=eta)
            ref = g[("DPM adaptive", order, eta)]
            assert {k: info[k] for k in ("steps", "n_accept", "n_reject")} == {k: ref["info"][k] for k in ("steps", "n_accept", "n_reject")}
            assert max_rel(got, ref["latent"]) < 2e-5, (order, eta)


def test_latent_resize_tables_match_torch_interpolate():
    """modules/latent_upscale.axis_table (host side of the hires-fix latent resize) applied as dense matrices == F.interpolate for every
    mode of shared.latent_upscale_modes, up- and down-scaling, odd sizes (the tie-breaking of nearest-exact needs fp32 index arithmetic)."""
    import torch.nn.functional as F
    from forge_amd.modules.latent_upscale import axis_table, latent_upscale_modes

    def apply(x, size, mode, aa):
        mats = []
        for n_in, n_out in ((x.shape[-2], size[0]), (x.shape[-1], size[1])):
            st, wt = axis_table(n_in, n_out, mode, aa)
            m = torch.zeros(n_out, n_in)
            for o in range(n_out):
                for k in range(wt.shape[1]):
                    m[o, int(st[o]) + k] += wt[o, k]
            mats.append(m)
        return torch.einsum("oh,bchw,pw->bcop", mats[0], x, mats[1])
    torch.manual_seed(0)
    sizes = (((16, 16), (32, 32)), ((8, 12), (20, 18)), ((64, 64), (96, 128)), ((16, 24), (24, 36)), ((13, 7), (29, 9)), ((32, 32), (24, 16)),
             ((20, 30), (7, 11)), ((16, 16), (16, 16)))
    for (h, w), out in sizes:
        x = torch.randn(2, 4, h, w)
        for m in latent_upscale_modes.values():
            kw = {"antialias": m["antialias"]} if m["mode"] in ("bilinear", "bicubic") else {}
            want = F.interpolate(x, size=out, mode=m["mode"], **kw)
            torch.testing.assert_close(apply(x, out, m["mode"], m["antialias"]), want, rtol=1e-5, atol=2e-5)


def test_hires_target_resolution_rules():
    from forge_amd.modules.processing import StableDiffusionProcessingTxt2Img as P
    def res(**kw):
        p = P(width=512, height=768, **kw)
        p.calculate_target_resolution()
        return p.hr_upscale_to_x, p.hr_upscale_to_y, p.truncate_x, p.truncate_y
    assert res(hr_scale=1.5) == (768, 1152, 0, 0)
    assert res(hr_resize_x=1024) == (1024, 1536, 0, 0)
    assert res(hr_resize_y=1024) == (682, 1024, 0, 0)
    assert res(hr_resize_x=1024, hr_resize_y=1024) == (1024, 1536, 0, 64)   # crop the long side: (1536 - 1024) // 8
    assert res(hr_resize_x=640, hr_resize_y=1280) == (853, 1280, 26, 0)


def test_prediction_types_and_beta_schedules():
    """backend/modules/k_prediction.py: all four beta schedules, zero-terminal-SNR rescale, percent_to_sigma, and calculate_denoised for
    v_prediction / edm -- oracle AND product (host tables) against the reference's Prediction class; the oracle's Euler run per prediction
    type against the reference stack."""
    from forge_amd.backend.modules import k_prediction as prod
    from oracle import k_prediction as okp
    g = load_golden("tiny_sd15_prediction_types.pt")
    for sched, (ls, le) in (("linear", (0.00085, 0.012)), ("cosine", (0.00085, 0.012)), ("sqrt_linear", (0.0001, 0.02)), ("sqrt", (0.0001, 0.0004))):
        want = g[("sigmas", sched)]
        assert torch.equal(okp.Predictor(ls, le, schedule=sched).sigmas, want), sched
        assert torch.equal(prod.Prediction(beta_schedule=sched, linear_start=ls, linear_end=le).sigmas, want), sched
    base = prod.Prediction(prediction_type="v_prediction")
    assert torch.equal(prod.rescale_zero_terminal_snr_sigmas(base.sigmas.clone()), g["ztsnr_sigmas"])
    assert torch.equal(okp.rescale_zero_terminal_snr_sigmas(okp.Predictor().sigmas.clone()), g["ztsnr_sigmas"])
    for pc, want in g["percent_to_sigma"].items():
        assert base.percent_to_sigma(pc) == want and okp.Predictor().percent_to_sigma(pc) == want, pc
    cfg = synth.TINY_SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    c, uc = synth.synth_conditioning(2, cfg["context_dim"], None, seed=1234)
    k = g["kat"]
    for ptype in ("v_prediction", "edm"):
        pred = okp.Predictor(prediction_type=ptype)
        torch.testing.assert_close(pred.calculate_denoised(k["sigma"], k["model_output"], k["x"]), g[("denoised", ptype)], rtol=1e-6, atol=1e-6)
        lat, _ = pipeline.txt2img_latents(sd, cfg, c, uc, g["seeds"], g["hw"] * 8, g["hw"] * 8, 4, sampler_name="Euler", predictor=pred)
        assert max_rel(lat, g[("euler4", ptype)]) < 2e-4, ptype
