"""CPU checks of the Flux rounding oracle (oracle/flux_sites.py) -- the checker behind tests/test_gpu_flux_sharp_parity.py.

Pin: without rounding it is oracle/flux.py's arithmetic (which tests/test_oracle_golden.py holds to the real reference's fixture), up to fp32 summation
order (the modulation Linear evaluated per chunk, LayerNorm statistics by another routine).  With rounding in bf16 / fp16 it must land where the reference's
own bf16 / fp16 run lands against its fp32 run (the floors)."""
import pytest
import torch

from forge_amd import synth
from oracle import flux as of
from oracle import flux_sites as fs

from conftest import load_golden
import parity


def _case():
    g = load_golden("tiny_flux_fwd.pt")
    cfg = synth.TINY_FLUX_CONFIG
    sd = synth.synth_flux_state_dict(cfg, seed=2)
    return g, cfg, sd


def _args(g):
    return g["x"], g["t"], g["ctx"], g["y"], g.get("guidance")


def test_without_rounding_it_is_the_pinned_restatement():
    g, cfg, sd = _case()
    x, t, ctx, y, gd = _args(g)
    a = of.flux_forward(sd, cfg, x, t, ctx, y, gd)
    b = fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=None)
    m = parity.metrics(b, a)
    print(m)
    assert m["max_rel"] < 2e-5
    assert parity.metrics(b, g["out"])["max_rel"] < 2e-4


@pytest.mark.parametrize("dt,tag", [(torch.bfloat16, "bf16"), (torch.float16, "f16")])
def test_with_rounding_it_sits_at_the_reference_floor(dt, tag):
    g, cfg, sd = _case()
    x, t, ctx, y, gd = _args(g)
    out = fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=dt)
    m = parity.metrics(out, g["out"])
    fl = parity.FLOORS[f"tiny_flux_fwd.pt:out@{tag}" if f"tiny_flux_fwd.pt:out@{tag}" in parity.FLOORS else "tiny_flux_fwd.pt:out"]
    print(tag, m, fl)
    assert 0.3 * fl["rms_rel"] <= m["rms_rel"] <= 1.4 * fl["rms_rel"]
    assert torch.equal(out, out.to(dt).float())


def test_teacher_forcing_with_its_own_outputs_reproduces_them():
    g, cfg, sd = _case()
    x, t, ctx, y, gd = _args(g)
    outs = {}
    a = fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=torch.bfloat16, layer_out=outs)
    outs2 = {}
    b = fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=torch.bfloat16, teacher=outs, layer_out=outs2)
    assert set(outs) == set(outs2) and len(outs) > 20
    assert torch.equal(a, b)
    for k in outs:
        assert torch.equal(outs[k], outs2[k]), k


def test_planted_bugs_show_at_their_stage():
    g, cfg, sd = _case()
    x, t, ctx, y, gd = _args(g)
    outs = {}
    fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=torch.float16, layer_out=outs)
    for plant, stage in (({"gelu_erf": "single_blocks.0"}, "single_blocks.0.mlp"), ({"gelu_erf": "double_blocks.0"}, "double_blocks.0.img.h")):
        bad = {}
        fs.flux_forward(sd, cfg, x, t, ctx, y, gd, dtype=torch.float16, teacher=outs, layer_out=bad, plant=plant)
        m = {k: parity.metrics(bad[k], outs[k])["rms_rel"] for k in outs}
        worst = max(m, key=m.get)
        print(plant, worst, m[worst])
        block = plant["gelu_erf"]
        hit = {k for k in m if k.startswith(block + ".") and k.endswith((".h", ".mlp"))}       # (a double block's plant sits in both streams' MLPs)
        assert stage in hit and worst in hit and m[stage] > 2.5e-4, (worst, m[worst], m[stage])   # (tanh vs erf GELU: ~3e-4 rms at these pre-activations)
        assert all(v < 1e-4 for k, v in m.items() if k not in hit)
