"""CPU checks of the executor-faithful rounding oracle (oracle/unet_fp16sites.py) -- the checker behind tests/test_gpu_sharp_parity.py.

Pin: with rounding off it IS the pinned fp32 restatement (bit for bit: the walk is oracle.unet's own, only leaf functions are swapped), which
tests/test_oracle_golden.py holds to the real reference's fixtures.  With rounding on it is an fp16 pipeline and must land where the reference's
own fp16 run lands against its fp32 run (the floor) -- the same order of error, not more.  The fold algebra and the planted bugs are checked to
have the size they are meant to have."""
import pytest
import torch

from forge_amd import synth
from oracle import unet as ou
from oracle import unet_fp16sites as o16
from oracle.make_golden import _inputs

from conftest import load_golden
import parity

TINY = {"tiny_sd15": synth.TINY_SD15_UNET_CONFIG, "tiny_sdxl": synth.TINY_SDXL_UNET_CONFIG}


def _case(name):
    cfg = TINY[name]
    g = load_golden(f"{name}_unet_fwd.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    return cfg, g, sd


def _blocks(sd):
    return sorted({k[:-len(".norm1.weight")] for k in sd if k.endswith(".norm1.weight")})


@pytest.mark.parametrize("name", list(TINY))
def test_rounding_off_is_the_pinned_restatement_bit_for_bit(name):
    cfg, g, sd = _case(name)
    a = ou.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"])
    b = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], rounding=False)
    assert torch.equal(a, b)
    # and the swap is undone: the pinned module's functions are its own again
    assert ou.resblock.__module__ == "oracle.unet" and ou._gn.__module__ == "oracle.unet" and ou.timestep_embedding.__module__ == "oracle.unet"
    assert torch.equal(ou.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"]), a)


@pytest.mark.parametrize("name", list(TINY))
def test_rounding_on_sits_at_the_reference_fp16_floor(name):
    cfg, g, sd = _case(name)
    eps = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"])
    m = parity.metrics(eps, g["eps"])
    fl = parity.FLOORS[f"{name}_unet_fwd.pt:eps"]
    print(name, m, fl)
    assert 0.4 * fl["rms_rel"] <= m["rms_rel"] <= 1.3 * fl["rms_rel"]
    assert torch.equal(eps, eps.half().float())         # what comes out is an fp16 tensor's values


@pytest.mark.parametrize("name", list(TINY))
def test_folded_layernorm_is_the_same_function_up_to_fp16_rounding(name):
    """all three folds on, every block: mathematically the same network; numerically other rounding sites -> differs from the unfolded run by
    about the floor's size, not more (a wrong fold formula would be off by O(1))"""
    cfg, g, sd = _case(name)
    fold = {b: (True, True, True) for b in _blocks(sd)}
    plain = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"])
    folded = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], fold=fold)
    m = parity.metrics(folded, plain)
    fl = parity.FLOORS[f"{name}_unet_fwd.pt:eps"]
    print(name, "folded vs plain", m)
    assert 0 < m["rms_rel"] <= 1.5 * fl["rms_rel"]
    assert parity.metrics(folded, g["eps"])["rms_rel"] <= 1.3 * fl["rms_rel"]


def _as_taps(outs):
    """layer outputs of an oracle run in the layout the executor's taps have (channels-last)"""
    return {k: (v.permute(0, 2, 3, 1).contiguous() if v.dim() == 4 else v) for k, v in outs.items()}


@pytest.mark.parametrize("name", list(TINY))
def test_two_correct_implementations_decorrelate_as_a_whole_and_agree_layer_by_layer(name):
    """The reason the sharp gate is layer-wise.  Implementation B = the same rounding network with fp64 instead of fp32 accumulation in every
    convolution and Linear (a perturbation of ~1e-7, standing in for another summation order).  Free-running, A and B end up in different
    rounding realisations: their outputs differ by most of the fp16 floor (> 0.35 x floor, the figure the round-4 verdict hoped a whole-network
    comparison could meet).  Layer by layer on shared inputs they agree to 1e-6 .. 6e-5 rms (median 8e-6) (a flipped fp16 rounding costs one element 1 ulp <= 9.8e-4): that comparison is sharp."""
    cfg, g, sd = _case(name)
    fl = parity.FLOORS[f"{name}_unet_fwd.pt:eps"]["rms_rel"]
    outs_a = {}
    a = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], layer_out=outs_a)
    b = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], acc64=True)
    whole = parity.metrics(b, a)["rms_rel"]
    outs_b = {}
    o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], acc64=True, teacher=_as_taps(outs_a), layer_out=outs_b)
    assert set(outs_a) == set(outs_b) and len(outs_a) > 10
    per_layer = {k: parity.metrics(outs_a[k], outs_b[k]) for k in outs_a}
    worst_rms = max(v["rms_rel"] for v in per_layer.values())
    worst_pp = max(v["pp_rel"] for v in per_layer.values())
    print(name, "whole network A vs B rms", whole, "=", whole / fl, "x floor; layer-wise worst rms", worst_rms, "pp", worst_pp)
    assert whole > 0.35 * fl
    assert worst_rms < 2e-4 and worst_pp < 1.5e-3     # measured 6e-5 / 1.1e-3


def test_teacher_forcing_with_its_own_outputs_is_the_identity():
    """... up to torch's CPU convolution not being invariant under the memory layout of its input (the free-running walk hands a permuted view
    from a SpatialTransformer to the next ResBlock, the teacher a contiguous tensor: another summation order, a handful of 1-ulp flips)"""
    cfg, g, sd = _case("tiny_sdxl")
    outs = {}
    a = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], layer_out=outs)
    outs2 = {}
    b = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], teacher=_as_taps(outs), layer_out=outs2)
    assert torch.equal(a, b)
    same = [k for k in outs if torch.equal(outs[k], outs2[k])]
    assert len(same) >= 0.85 * len(outs)
    for k in outs:
        m = parity.metrics(outs2[k], outs[k])
        assert m["rms_rel"] < 2e-4 and m["pp_rel"] < 1.5e-3, (k, m)


def test_planted_bugs_are_visible_layer_wise_and_mostly_invisible_to_the_floor_gate():
    """the two planted constants (GroupNorm eps 1e-6 in one ResBlock norm whose input is small; tanh GELU in one GEGLU), evaluated as the GPU
    test does but with the rounding oracle itself as the "native" side: the planted layer -- and only it -- exceeds the sharp gate"""
    from test_gpu_sharp_parity import PLANTS, over_gate, planted_layer, small_variance_state_dict
    cfg = TINY["tiny_sd15"]
    g = load_golden("tiny_sd15_unet_fwd.pt")
    for pname, plant in PLANTS.items():
        sd = small_variance_state_dict(cfg) if "gn_eps" in plant else synth.synth_unet_state_dict(cfg, seed=0)
        outs = {}
        o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], layer_out=outs)
        bad = {}
        o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], plant=plant, teacher=_as_taps(outs), layer_out=bad)
        m = {k: parity.metrics(outs[k], bad[k]) for k in outs}
        failing = over_gate(m)
        layer = planted_layer(plant)
        print(pname, layer, m[layer], failing)
        assert failing == [layer], (pname, failing, m[layer])


def test_planted_bugs_inside_the_layernorm_fold_are_visible_at_their_layer():
    """The two fold plants of tests/test_gpu_sharp_parity.py (LayerNorm eps 1e-6 inside a folded norm2 of a small-variance stream; colsum(W) for
    colsum(W * gamma) in a folded ff.net.0), with the rounding oracle itself as the "native" side and every norm folded, through the single-block
    evaluator the GPU test uses at full size: the planted layer -- and only it -- exceeds the sharp gate; unplanted, the block agrees with the
    whole-network walk's own layers."""
    from test_gpu_sharp_parity import fold_planted_layer, fold_plants, over_gate, small_stream_state_dict
    cfg, g, sd0 = _case("tiny_sdxl")
    block = _blocks(sd0)[0]
    sd = small_stream_state_dict(sd0, block)
    fold = {b: (True, True, True) for b in _blocks(sd)}
    outs = {}
    o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], fold=fold, layer_out=outs)
    taps = _as_taps(outs)
    same = {}
    o16.transformer_block_only(sd, cfg, block, g["ctx"], fold, taps, layer_out=same)
    assert len(same) == 12
    for k, v in same.items():
        m = parity.metrics(v, outs[k])
        assert m["rms_rel"] < 2e-4 and m["pp_rel"] < 1.5e-3, (k, m)
    for pname, plant in fold_plants(block).items():
        bad = {}
        o16.transformer_block_only(sd, cfg, block, g["ctx"], fold, taps, plant=plant, layer_out=bad)
        m = {k: parity.metrics(outs[k], bad[k]) for k in bad}
        failing = over_gate(m)
        layer = fold_planted_layer(plant)
        print(pname, layer, m[layer], failing)
        assert failing == [layer], (pname, failing, m[layer])


def test_upsample_conv_as_four_phase_convolutions_is_the_same_function():
    """The Upsample convolution on tap-summed weights (fmx_conv3x3_up2x; oracle site `up2x`): with rounding off it IS interpolate + conv3x3 (fp32
    summation order apart); the native weight fold (forge_amd.hipops.fold_up2x_weights: [4][nout][2][2][c]) holds exactly the oracle's tap sums; with
    rounding on, the extra rounding of the sums moves the network output by a fraction of the fp16 floor."""
    import torch.nn.functional as F
    from forge_amd import hipops
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(2, 6, 5, 7, generator=g), torch.randn(4, 6, 3, 3, generator=g), torch.randn(4, generator=g)
    xu = F.interpolate(x, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xu, w, b, padding=1)
    torch.testing.assert_close(o16.up2x_phase_conv(xu, w, b, lambda t: t), ref, rtol=1e-5, atol=2e-5)
    w4 = hipops.fold_up2x_weights(w.permute(0, 2, 3, 1).reshape(4, -1).contiguous(), 6)
    out = torch.empty_like(ref)
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        out[:, :, py::2, px::2] = F.conv2d(F.pad(x, (1 - px, px, 1 - py, py)), w4[ph].view(4, 2, 2, 6).permute(0, 3, 1, 2), b)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)
    cfg, gg, sd = _case("tiny_sdxl")
    ups = sorted({k[:-len(".conv.weight")] for k in sd if k.endswith(".conv.weight")})
    assert ups
    plain = o16.unet_forward(sd, cfg, gg["x"], gg["t"], gg["ctx"], gg["y"])
    phased = o16.unet_forward(sd, cfg, gg["x"], gg["t"], gg["ctx"], gg["y"], up2x=ups)
    m = parity.metrics(phased, plain)
    fl = parity.FLOORS["tiny_sdxl_unet_fwd.pt:eps"]
    print("up2x vs plain", m, "floor", fl["rms_rel"])
    assert 0 < m["rms_rel"] <= 1.0 * fl["rms_rel"]
    assert parity.metrics(phased, gg["eps"])["rms_rel"] <= 1.3 * fl["rms_rel"]
    assert torch.equal(o16.unet_forward(sd, cfg, gg["x"], gg["t"], gg["ctx"], gg["y"], rounding=False, up2x=ups),
                       o16.unet_forward(sd, cfg, gg["x"], gg["t"], gg["ctx"], gg["y"], rounding=False))
