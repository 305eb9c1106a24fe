"""CPU checks of the parity bookkeeping (tests/parity.py) and of the committed fp16 floors (tests/golden/fp16_floor.json): every floor key the
GPU tests look up exists (a typo must fail here, not on the GPU box), the floors are in the range an fp16 pipeline can have, the metric
definitions agree between oracle/make_floor.py and tests/parity.py, and -- when /root/reference is present -- one floor is regenerated
from the real reference and compared with the committed value."""
import pytest
import torch

import parity
from oracle import ref_import

EXTRA = ["Heun", "DPM2", "DPM2 a", "DPM++ 2S a", "LMS", "HeunPP2", "IPNDM", "IPNDM_V", "DEIS", "Restart"]
MORE = ["DDIM", "DDIM eta", "DDIM CFG++", "PLMS", "LCM", "DDPM"]


def expected_keys():
    keys = []
    for name in ("tiny_sd15", "tiny_sdxl"):
        keys += [f"{name}_unet_fwd.pt:eps", f"{name}_unet_hooks.pt:eps", f"{name}_unet_hooks.pt:euler3/latent", f"{name}_controlnet.pt:euler4/latent"]
        keys += [f"{name}_samples.pt:{s}/latent" for s in ("Euler", "Euler a", "DPM++ 2M", "Euler_cfg1")]
        keys += [f"{name}_cfg_paths.pt:{k}" for k in ("plain", "and_composed", "cfg_functions", "model_function_wrapper")]
    keys += [f"tiny_sd15_samples_extra.pt:{s}/latent" for s in EXTRA]
    keys += [f"tiny_sd15_samples_more.pt:{s}/latent" for s in MORE]
    keys += [f"tiny_sd15_samples_unipc.pt:{n}/latent" for n in (6, 9)]
    keys += [f"samplers_sde_dpm.pt:stack/{s}/latent" for s in ("DPM++ SDE", "DPM++ 2M SDE", "DPM++ 3M SDE", "DPM fast", "DPM adaptive")]
    keys += [f"tiny_sd15_img2img.pt:{s}/latent" for s in ("Euler", "Euler a", "DPM++ 2M", "Euler_masked")]
    keys += [f"tiny_sd15_prediction_types.pt:('euler4', '{p}')" for p in ("v_prediction", "edm")]
    keys += ["tiny_sd15_unet_ctrl.pt:eps", "tiny_sd15_inpaint_model.pt:eps", "tiny_sd15_inpaint_model.pt:euler3", "mini_sd15_t2i_adapter.pt:euler3",
             "tiny_vae_decode.pt:decode", "tiny_vae_decode.pt:decode_first_stage", "tiny_flux_vae_decode.pt:decode",
             "tiny_flux_vae_decode.pt:decode_first_stage", "tiny_vae_encode.pt:moments", "tiny_vae_encode.pt:sample",
             "pipeline:txt2img_eulera4/latent", "pipeline:txt2img_eulera4/decoded", "pipeline:smoke_euler3/latent", "pipeline:smoke_euler3/decoded",
             "sd15_config0.pt:eps", "sd15_config0.pt:latent", "sd15_config0.pt:decoded", "sdxl_vae1024.pt:decoded"]
    # the rows next to the hot path (oracle/make_floor.py floors_aux): ControlNet residuals, adapters, text encoders, Flux in both compute types
    keys += ["tiny_sd15_controlnet.pt:outs_worst", "tiny_sdxl_controlnet.pt:outs_worst", "mini_adapter_light.pt:features_worst",
             "tiny_sd15_unet_module_hooks.pt:eps", "tiny_sdxl_unet_module_hooks.pt:eps"]
    keys += [f"mini_sd15_t2i_adapter.pt:features/{v}_worst" for v in ("sd15_k1_pool", "sd15_k3_conv", "sdxl")]
    keys += [f"tiny_clip_{e}.pt:{k}" for e in "lg" for k in ("last_hidden_state", "hidden_penultimate", "penultimate_final_ln", "pooled")]
    keys += ["tiny_clip_g.pt:pooled_projected"] + [f"tiny_flux_fwd.pt:{k}@{t}" for k in ("out", "latent") for t in ("f16", "bf16")]
    return keys


def test_every_floor_the_gpu_tests_use_is_committed():
    missing = [k for k in expected_keys() if k not in parity.FLOORS]
    assert not missing, missing


def test_full_size_fixtures_come_with_their_floor():
    import os
    from conftest import GOLDEN
    for fixture, key in (("sd15_config2.pt", "sd15_config2.pt:latent"), ("sdxl_full_fwd.pt", "sdxl_full_fwd.pt:eps"),
                         ("sdxl_config3.pt", "sdxl_config3.pt:latent"), ("sdxl_config3_decode.pt", "sdxl_config3_decode.pt:decoded")):
        if os.path.exists(os.path.join(GOLDEN, fixture)):
            assert key in parity.FLOORS, f"{fixture} is committed without its fp16 floor {key}"


def test_floors_are_fp16_sized():
    """A floor is the reference's own fp16-vs-fp32 discrepancy: above fp16 epsilon (4.9e-4) in the max norm for anything that passed through a
    network, and far below a percent for one forward / one decode."""
    for k, v in parity.FLOORS.items():
        hi = 1e-1 if k.endswith("@bf16") else 2e-2   # bfloat16 (the reference's Flux compute type) has 8 significand bits
        if k.startswith("tiny_t5.pt:"):
            # the reference's T5 squares and averages its RMS-norm statistic IN the 16-bit type (backend/nn/t5.py:21-23) on an un-normalised residual
            # stream: its own 16-bit run is 2.9e-2 (fp16) / 1.2e-1 (bf16) off its fp32 run -- the native encoder, with fp32 statistics, sits well inside
            hi = 2e-1 if k.endswith("@bf16") else 5e-2
        assert 2e-4 < v["max_rel"] < hi and v["rms_rel"] <= v["pp_rel"] and v["max_rel"] <= v["pp_rel"] * 1.0001, (k, v)
    assert parity.FLOORS["tiny_sd15_unet_fwd.pt:eps"]["max_rel"] < 4e-3
    assert parity.FLOORS["sd15_config0.pt:latent"]["max_rel"] < 4e-3


def test_metric_definitions_and_limits():
    from oracle.make_floor import metrics as floor_metrics
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(2, 4, 16, 16, generator=g)
    got = ref + 1e-3 * torch.randn(2, 4, 16, 16, generator=g)
    a, b = parity.metrics(got, ref), floor_metrics(got, ref)
    for k in a:
        assert abs(a[k] - b[k]) < 1e-12
    d = (got - ref).abs()
    assert abs(a["max_rel"] - float(d.max() / ref.abs().max())) < 1e-9
    rms = float(ref.pow(2).mean().sqrt())
    assert abs(a["pp_rel"] - float((d / ref.abs().clamp_min(rms)).max())) < 1e-9
    fl, lim = parity.limits("tiny_sd15_unet_fwd.pt:eps")
    assert lim["max_rel"] == max(1e-3, 1.5 * fl["max_rel"]) and lim["rms_rel"] == max(1e-3, 1.25 * fl["rms_rel"])
    fl2, _ = parity.limits(["tiny_sd15_unet_fwd.pt:eps", "tiny_sdxl_unet_fwd.pt:eps"])
    assert fl2["max_rel"] == max(parity.FLOORS["tiny_sd15_unet_fwd.pt:eps"]["max_rel"], parity.FLOORS["tiny_sdxl_unet_fwd.pt:eps"]["max_rel"])
    with pytest.raises(AssertionError):
        parity.check("way off", ref * 1.1, ref, floor="tiny_sd15_unet_fwd.pt:eps")
    parity.check("exact", ref, ref, floor="tiny_sd15_unet_fwd.pt:eps")
    pair, _ = parity.limits("tiny_sd15_unet_fwd.pt:eps", both_fp16=True)
    assert abs(pair["max_rel"] - fl["max_rel"] * 2 ** 0.5) < 1e-12


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference")
def test_committed_floor_is_reproduced_by_the_real_reference_in_fp16():
    from conftest import load_golden
    from forge_amd import synth
    from oracle.make_floor import half_unet, metrics
    cfg = synth.TINY_SD15_UNET_CONFIG
    g = load_golden("tiny_sd15_unet_fwd.pt")
    with torch.no_grad():
        eps16 = half_unet(cfg)(g["x"], g["t"], context=g["ctx"], y=None, transformer_options={})
    m = metrics(eps16, g["eps"])
    want = parity.FLOORS["tiny_sd15_unet_fwd.pt:eps"]
    assert abs(m["max_rel"] - want["max_rel"]) < 0.05 * want["max_rel"], (m, want)
