"""SHARP parity of the VAE decode: the native decoder against the rounding oracle oracle/vae_fp16sites.py -- the pinned restatement of
backend/nn/vae.py:248-271 with fp16 rounding at exactly the executor's storage sites -- LAYER BY LAYER, teacher-forced, as tests/test_gpu_sharp_parity.py does
for the UNet (why layer-wise: DESIGN.md 2.4).  The executor hands out every tensor it stores (`IntegratedAutoencoderKL.tap`: conv_in, every ResnetBlock's conv1
output and output, q / k / V^T / attention output / AttnBlock output of the mid attention, every Upsample convolution, conv_out), the oracle evaluates each
layer on the executor's own stored input of that layer, and every layer is gated like the UNet's: rms <= 2e-4 (attention output 5e-4: the fused kernel rounds
P at the running-max scale), per-pixel <= 3.5e-3.  The 1024^2 decode's END-to-end figure against the fp32 reference is 7e-3 .. 2e-2 per pixel
(the fp16 floor, tests/test_gpu_e2e.py); layer by layer against the same arithmetic at the same storage precision it is three orders of magnitude tighter.
A planted bug (GroupNorm eps 1e-5 for 1e-6 in one norm, vae.py:12-13) must fail at exactly its layer."""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402,F401
from forge_amd import synth  # noqa: E402
from forge_amd.backend.nn.vae import IntegratedAutoencoderKL  # noqa: E402
from oracle import vae_fp16sites as v16  # noqa: E402

import parity  # noqa: E402

DEV = "cuda"
# per pixel 3.5e-3 where the UNet's gate is 2.5e-3: the decoder's tensors have up to 1.3e8 elements (128 channels x 1024^2) -- the extreme value of that many
# one- and two-ulp flips was measured at 2.4e-3 (profiles/r35_vae_sharp_parity.jsonl); rms gates as for the UNet
SHARP_RMS, SHARP_RMS_ATTN, SHARP_PP = 2.0e-4, 5.0e-4, 3.5e-3


def _gate(key):
    return SHARP_RMS_ATTN if key.endswith((".o", "attn_1")) else SHARP_RMS


def _log(rec):
    print("[sharp-vae]", json.dumps(rec))
    path = os.environ.get("FMX_SHARP_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def _layerwise(cfg, z, plant=None, seed=1, scale_conv_in=None):
    sd = synth.synth_vae_decoder_state_dict(cfg, seed=seed)
    if scale_conv_in is not None:     # a small-variance stream in front of the first ResnetBlock: there the GroupNorm epsilon matters
        sd = {k: (v * scale_conv_in if k.startswith("decoder.conv_in.") else v) for k, v in sd.items()}
    vae = IntegratedAutoencoderKL(cfg, sd, device=DEV, dtype=torch.float16, auto_bf16_fallback=False)
    vae.tap = {}
    out = vae.decode(z.to(DEV)).float().cpu()
    taps, vae.tap = vae.tap, None
    assert torch.isfinite(out).all()
    t0 = time.time()
    mine = {}
    v16.vae_decode(sd, z, teacher=taps, layer_out=mine, plant=plant, up2x=vae.up2x_trace)   # (which Upsample convolutions ran as four phase convolutions)
    secs = time.time() - t0
    # (the materialised attention of widths other than 512 computes V^T image by image inside its loop: no tap for it)
    assert set(taps) <= set(mine) and set(mine) - set(taps) <= {"decoder.mid.attn_1.v"}, sorted(set(mine) ^ set(taps))
    torch.testing.assert_close(taps["conv_out"], out, rtol=0, atol=0)      # what decode() returns IS the last tap
    return {k: parity.metrics(taps[k], mine[k]) for k in taps}, secs, vae


def _summary(name, m, secs):
    by = {}
    for k, v in m.items():
        b = by.setdefault(v16.kind_of(k), {"layers": 0, "worst_rms_rel": 0.0, "worst_pp_rel": 0.0})
        b["layers"] += 1
        b["worst_rms_rel"] = max(b["worst_rms_rel"], round(v["rms_rel"], 8))
        b["worst_pp_rel"] = max(b["worst_pp_rel"], round(v["pp_rel"], 8))
    rms = sorted(v["rms_rel"] for v in m.values())
    wr, wp = max(m, key=lambda k: m[k]["rms_rel"]), max(m, key=lambda k: m[k]["pp_rel"])
    _log({"name": name, "layers": len(m), "worst_rms_rel": round(m[wr]["rms_rel"], 8), "worst_rms_layer": wr, "worst_pp_rel": round(m[wp]["pp_rel"], 8),
          "worst_pp_layer": wp, "median_rms_rel": round(rms[len(rms) // 2], 8), "gate": {"rms_rel": SHARP_RMS, "rms_rel_attention": SHARP_RMS_ATTN, "pp_rel": SHARP_PP},
          "oracle_seconds": round(secs, 1), "by_kind": by})


def _over(m):
    return sorted(k for k, v in m.items() if v["rms_rel"] > _gate(k) or v["pp_rel"] > SHARP_PP)


def test_tiny_vae_layer_by_layer():
    g = torch.Generator("cpu").manual_seed(5)
    z = torch.randn(2, synth.TINY_VAE_CONFIG["latent_channels"], 16, 24, generator=g)
    m, secs, _ = _layerwise(synth.TINY_VAE_CONFIG, z)
    _summary("sharp layer-wise VAE decode: tiny VAE (materialised attention)", m, secs)
    assert not _over(m), {k: m[k] for k in _over(m)}


@pytest.mark.parametrize("lat", [64, 128])
def test_sdxl_vae_layer_by_layer(lat):
    """the SDXL decoder at 512^2 and at 1024^2 (the bench's decode: fused 512-wide attention over 4 096 / 16 384 tokens, 512x128 tiles, the direct conv_out)"""
    g = torch.Generator("cpu").manual_seed(7)
    z = torch.randn(1, 4, lat, lat, generator=g) * 0.9
    m, secs, vae = _layerwise(synth.SDXL_VAE_CONFIG, z)
    from forge_amd import hipops as _ops
    if _ops._UP2X:
        assert len(vae.up2x_trace) == (3 if lat == 128 else 2), vae.up2x_trace      # the Upsample convolutions whose phases fill the chip (>= 128 tiles each) run as phase convolutions
    _summary(f"sharp layer-wise VAE decode: SDXL VAE, {8 * lat}^2 ({len(vae.up2x_trace)} Upsample convolutions as phase convolutions)", m, secs)
    assert not _over(m), {k: m[k] for k in _over(m)}


def test_a_planted_groupnorm_eps_fails_at_its_layer_and_nowhere_else():
    g = torch.Generator("cpu").manual_seed(5)
    z = torch.randn(2, synth.TINY_VAE_CONFIG["latent_channels"], 16, 24, generator=g)
    key = "decoder.mid.block_1"
    m, secs, _ = _layerwise(synth.TINY_VAE_CONFIG, z, plant={"gn_eps": (key, "norm1", 1e-5)}, scale_conv_in=0.05)
    _log({"name": "planted bug: GroupNorm eps 1e-5 instead of 1e-6 in one ResnetBlock norm of the VAE (vae.py:12-13)", "layer": key + ".h",
          "sharp_gate_fails_at": _over(m), "planted_layer_rms_rel": round(m[key + ".h"]["rms_rel"], 7)})
    assert _over(m) == [key + ".h"], _over(m)
    clean, _, _ = _layerwise(synth.TINY_VAE_CONFIG, z, scale_conv_in=0.05)
    assert not _over(clean), _over(clean)
