"""CPU checks of the VAE rounding oracle (oracle/vae_fp16sites.py) -- the checker behind tests/test_gpu_vae_sharp_parity.py.

Pin: with rounding off it IS oracle/vae.py's walk (bit for bit), which tests/test_oracle_golden.py holds to the real reference's fixtures.  With rounding
on it is an fp16 pipeline and must land where the reference's own fp16 decode lands against its fp32 decode (the floor)."""
import pytest
import torch

from forge_amd import synth
from oracle import vae as ov
from oracle import vae_fp16sites as v16

from conftest import load_golden
import parity


def _case():
    g = load_golden("tiny_vae_decode.pt")
    sd = synth.synth_vae_decoder_state_dict(synth.TINY_VAE_CONFIG, seed=1)
    return g, sd


def test_rounding_off_is_the_pinned_restatement_bit_for_bit():
    g, sd = _case()
    z = g["z"] if "z" in g else g["latent"]
    a = ov.vae_decode(sd, z.float())
    b = v16.vae_decode(sd, z, rounding=False)
    assert torch.equal(a, b)


def test_rounding_on_sits_at_the_reference_fp16_floor():
    g, sd = _case()
    z = g["z"] if "z" in g else g["latent"]
    want = ov.vae_decode(sd, z.float())
    got = v16.vae_decode(sd, z)
    m = parity.metrics(got, want)
    fl = next(v for k, v in parity.FLOORS.items() if k.startswith("tiny_vae_decode.pt:"))
    print(m, fl)
    assert 0.3 * fl["rms_rel"] <= m["rms_rel"] <= 1.3 * fl["rms_rel"]
    assert torch.equal(got, got.half().float())


def test_teacher_forcing_with_its_own_outputs_reproduces_them():
    g, sd = _case()
    z = g["z"] if "z" in g else g["latent"]
    outs = {}
    a = v16.vae_decode(sd, z, layer_out=outs)
    outs2 = {}
    b = v16.vae_decode(sd, z, teacher=outs, layer_out=outs2)
    assert set(outs) == set(outs2) and len(outs) >= 10
    for k in outs:
        m = parity.metrics(outs2[k], outs[k])
        assert m["rms_rel"] < 2e-4 and m["pp_rel"] < 1.5e-3, (k, m)
    assert parity.metrics(b, a)["rms_rel"] < 2e-4


def test_a_planted_groupnorm_eps_is_visible_at_its_layer_only():
    g, sd = _case()
    z = g["z"] if "z" in g else g["latent"]
    sd = {k: (v * 0.05 if k.endswith("decoder.conv_in.weight") or k.endswith("decoder.conv_in.bias") else v) for k, v in sd.items()}   # a small-variance stream: eps matters
    outs = {}
    v16.vae_decode(sd, z, layer_out=outs)
    bad = {}
    key = "decoder.mid.block_1"
    v16.vae_decode(sd, z, teacher=outs, layer_out=bad, plant={"gn_eps": (key, "norm1", 1e-5)})
    m = {k: parity.metrics(bad[k], outs[k]) for k in outs}
    worst = max(m, key=lambda k: m[k]["rms_rel"])
    print(worst, m[worst])
    assert worst == key + ".h" and m[worst]["rms_rel"] > 5e-4
    assert all(v["rms_rel"] < 2e-4 for k, v in m.items() if k != key + ".h")
