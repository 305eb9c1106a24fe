"""GPU (MI355X): LoRA merge on the device GEMM and checkpoint loading through forge_amd.backend.loader, against the merged
weights produced by the REAL reference (tests/golden/tiny_sd15_lora_merge.pt) and the CPU oracle run on those weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402
from forge_amd import synth  # noqa: E402
from forge_amd.backend import loader  # noqa: E402
from forge_amd.backend.patcher import lora as nlora  # noqa: E402
from oracle.make_golden import synth_lora  # noqa: E402
from oracle.unet import unet_forward  # noqa: E402

from conftest import load_golden  # noqa: E402
from parity import check  # noqa: E402

DEV = "cuda"
CFG = synth.TINY_SD15_UNET_CONFIG


def max_rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max())


def test_lora_merge_on_device_vs_reference():
    g = load_golden("tiny_sd15_lora_merge.pt")
    sd = {k: v.half() for k, v in synth.synth_unet_state_dict(CFG, seed=0).items()}
    merged, report = nlora.merge_loras_into_state_dict(sd, CFG, [(synth_lora(CFG), g["strength"])], device=DEV)
    assert report["patched"] == len(g["merged"])
    for k, ref in g["merged"].items():
        got = merged[k].float().cpu().reshape(ref.shape)
        err = (got - ref.float()).abs().max().item()
        # same fp32 accumulation, one fp16 rounding on both sides: at most 1 fp16 ulp of the largest weight
        assert err <= 2.0 ** -10 * ref.float().abs().max().item() + 1e-6, (k, err)
        print(f"[parity] lora merge {k}: max abs err {err:.3e} (ref max {ref.float().abs().max().item():.3f})")
    # untouched tensors are passed through
    assert merged["time_embed.0.weight"] is sd["time_embed.0.weight"]


def test_stacked_loras_on_one_key_round_once():
    """Three patches on the same weight (two LoRAs and a full diff): the reference casts the weight to fp32, applies all of them and rounds
    once (backend/patcher/lora.py:85-92, :322).  The native merge keeps an fp32 running weight between patches; against that arithmetic in
    torch fp32 the result is within HALF an fp16 ulp of the exact value, i.e. at most one ulp from the rounded reference -- rounding after
    every patch (round 1's behaviour) drifts by up to the number of patches."""
    torch.manual_seed(3)
    out_f, in_f, r1, r2 = 320, 192, 8, 16
    w = (torch.randn(out_f, in_f) * 0.05).half()
    up1, dn1 = torch.randn(out_f, r1) * 0.1, torch.randn(r1, in_f) * 0.1
    up2, dn2 = torch.randn(out_f, r2) * 0.1, torch.randn(r2, in_f) * 0.1
    diff = torch.randn(out_f, in_f) * 0.01
    patches = [(0.8, ("lora", (up1, dn1, 4.0, None, None)), 1.0, None, None),
               (-0.5, ("lora", (up2, dn2, None, None, None)), 1.0, None, None),
               (1.0, ("diff", (diff,)), 1.0, None, None)]
    got = nlora.merge_lora_to_weight(patches, w, key="stacked", device=DEV).float().cpu()
    f16 = lambda t: t.half().double()   # the LoRA factors enter the device GEMM as fp16
    exact = w.double() + 0.8 * (4.0 / r1) * (f16(up1) @ f16(dn1)) - 0.5 * (f16(up2) @ f16(dn2)) + diff.double()
    ulp = torch.maximum(exact.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -10
    err = (got.double() - exact).abs()
    assert bool((err <= 0.5 * ulp * 1.01 + 1e-7).all()), f"max error {float((err / ulp).max()):.3f} ulp (0.5 = one correct rounding)"
    single = nlora.merge_lora_to_weight(patches[:1], w, key="single", device=DEV).float().cpu()
    exact1 = w.double() + 0.8 * (4.0 / r1) * (f16(up1) @ f16(dn1))
    ulp1 = torch.maximum(exact1.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -10
    assert bool(((single.double() - exact1).abs() <= 0.5 * ulp1 * 1.01 + 1e-7).all()), "single patch: fused fp16-residual GEMM, one rounding"


def test_forge_loader_with_lora_end_to_end():
    """single-file checkpoint layout (model.diffusion_model.* + first_stage_model.*) -> engine with a merged LoRA; the UNet
    forward must match the CPU oracle evaluated on the reference-merged weights."""
    g = load_golden("tiny_sd15_lora_merge.pt")
    base = synth.synth_unet_state_dict(CFG, seed=0)
    ckpt = {loader.UNET_PREFIX + k: v.half() for k, v in base.items()}
    parts = {"unet": {k: v.half() for k, v in base.items()}}
    merged, _ = nlora.merge_loras_into_state_dict(parts["unet"], CFG, [(synth_lora(CFG), g["strength"])], device=DEV)
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    net = IntegratedUNet2DConditionModel(CFG, merged, device=DEV)
    ref_sd = {k: v.float() for k, v in base.items()}
    ref_sd = {k: v.half().float() for k, v in ref_sd.items()}
    for k, v in g["merged"].items():
        ref_sd[k] = v.float()
    fx = load_golden("tiny_sd15_unet_fwd.pt")
    eps = net.forward(fx["x"].to(DEV), fx["t"].to(DEV), context=fx["ctx"].to(DEV), y=None)
    want = unet_forward(ref_sd, CFG, fx["x"], fx["t"], fx["ctx"], None)
    base_out = fx["eps"]
    # the same network and inputs as the plain forward fixture, with merged weights: that fixture's floor
    check(f"UNet forward with merged LoRA vs oracle on reference-merged weights (LoRA moved the output by {max_rel(want, base_out):.3e})", eps, want,
          floor="tiny_sd15_unet_fwd.pt:eps")
    assert max_rel(want, base_out) > 1e-2, "the synthetic LoRA must change the output measurably"
    # the split / detection entry point on the same checkpoint (tiny config has no family rule -> detection is shape-only)
    unet_part = {k[len(loader.UNET_PREFIX):]: v for k, v in loader.preprocess_state_dict(ckpt).items()}
    assert set(unet_part) == set(base)


def test_flux_lora_merge_on_device_vs_reference():
    """A diffusers-named Flux LoRA merged into the tiny Flux transformer on the device -- slices of the fused qkv / linear1 projections patched one by
    one, norm_out through swap_scale_shift, native and OneTrainer spellings beside it -- against the REAL reference's merge (tests/golden/tiny_flux_lora_merge.pt)."""
    from oracle.make_golden import synth_flux_lora
    g = load_golden("tiny_flux_lora_merge.pt")
    cfg = synth.TINY_FLUX_CONFIG
    sd = {k: v.half() for k, v in synth.synth_flux_state_dict(cfg, seed=2).items()}
    merged, report = nlora.merge_loras_into_flux_state_dict(sd, cfg, [(synth_flux_lora(cfg), g["strength"])], device=DEV, dtype=torch.float16)
    assert report["patched"] == len(g["merged"]) and report["unused_keys"] == [[]]
    for k, ref in g["merged"].items():
        got = merged[k].float().cpu().reshape(ref.shape)
        err = (got - ref.float()).abs().max().item()
        assert err <= 2.0 ** -10 * ref.float().abs().max().item() + 1e-6, (k, err)          # one fp16 rounding on both sides
        assert not torch.equal(merged[k].cpu(), sd[k])
    hs = cfg["hidden_size"]
    qkv, q0 = merged["double_blocks.0.txt_attn.qkv.weight"].cpu(), sd["double_blocks.0.txt_attn.qkv.weight"]
    assert torch.equal(qkv[:hs], q0[:hs]) and torch.equal(qkv[2 * hs:], q0[2 * hs:]) and not torch.equal(qkv[hs:2 * hs], q0[hs:2 * hs])
    assert merged["img_in.weight"] is sd["img_in.weight"]


def test_flux_lora_merge_on_bfloat16_storage_vs_reference():
    """The same LoRA on Flux's own storage type: the reference casts the bf16 weight to fp32, merges, casts ONCE to bf16 (patcher/lora.py:85-92,
    :322) -- so does the native merge (no detour through fp16).  Fixture `merged_bf16` from the real reference."""
    from oracle.make_golden import synth_flux_lora
    g = load_golden("tiny_flux_lora_merge.pt")
    cfg = synth.TINY_FLUX_CONFIG
    sd = {k: v.bfloat16() for k, v in synth.synth_flux_state_dict(cfg, seed=2).items()}
    merged, report = nlora.merge_loras_into_flux_state_dict(sd, cfg, [(synth_flux_lora(cfg), g["strength"])], device=DEV, dtype=torch.bfloat16)
    assert report["patched"] == len(g["merged_bf16"])
    for k, ref in g["merged_bf16"].items():
        assert merged[k].dtype == torch.bfloat16
        got, ref = merged[k].float().cpu().reshape(ref.shape), ref.float()
        # the LoRA factors enter the device GEMM as fp16 (the reference multiplies them in fp32: the delta carries <= 2 x 2^-11 of itself), then ONE bf16
        # rounding on both sides: one ulp of the result plus that share of the largest delta
        delta = float((ref - sd[k].float().reshape(ref.shape)).abs().max())
        assert bool(((got - ref).abs() <= 2.0 ** -7 * ref.abs() + 1.0e-3 * delta).all()), k
        assert (got != ref).float().mean().item() < 0.10, (k, (got != ref).float().mean().item())
    hs = cfg["hidden_size"]
    q0 = sd["double_blocks.0.txt_attn.qkv.weight"]
    assert torch.equal(merged["double_blocks.0.txt_attn.qkv.weight"].cpu()[:hs], q0[:hs])           # the un-patched slice: bit for bit
    # what a detour through fp16 would lose: bf16 values below fp16's range survive a merge that does not touch them (a zero diff), above it they do not overflow
    w = torch.full((64, 64), 2.0 ** -20, dtype=torch.bfloat16)
    w[0, 0] = 1.0e5
    out = nlora.merge_lora_to_weight([(1.0, ("diff", (torch.zeros(64, 64),)), 1.0, None, None)], w, device=DEV, out_dtype=torch.bfloat16)
    assert out.dtype == torch.bfloat16 and torch.equal(out.cpu(), w)
