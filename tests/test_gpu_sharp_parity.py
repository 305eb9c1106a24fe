"""SHARP parity (VERDICT r4 "Next round" item 2): the native UNet against the executor-faithful rounding oracle (oracle/unet_fp16sites.py: the
pinned fp32 restatement of backend/nn/unet.py:696-763 with fp16 rounding at exactly the executor's storage sites).

Two granularities:

* LAYER-WISE, teacher-forced (the gate).  The executor hands out every tensor it stores (`IntegratedUNet2DConditionModel.tap`: conv_in, the
  time / label embedding, every ResBlock's conv1 + emb and output, proj_in, and per BasicTransformerBlock the q / k / v operands of both attention
  launches, both attention outputs, the stream after attn1 / attn2 / the feed-forward and the GEGLU output, every SpatialTransformer's output, Down,
  Up, the output head: 124 tensors in the tiny networks, 277 in SD1.5, 903 in SDXL); the oracle evaluates each segment on the executor's OWN stored
  inputs and the two results are compared.  Both sides round at the same places from the same inputs, so what is left is fp32 summation order
  plus the rare fp16 rounding flip it causes (one element off by 1 ulp <= 9.8e-4): two CPU implementations that differ only in accumulation
  precision agree to 1e-6 .. 6e-5 rms this way (tests/test_oracle_fp16sites.py).  Measured on MI355X (profiles/r22_sharp_parity.jsonl): median
  1.2e-5 .. 2.3e-5 rms per tensor, every tensor that is not an attention output <= 5e-5 (time embedding 1.9e-4: 1280 numbers through four rounding
  levels), attention outputs 2e-5 .. 3.4e-4 (see SHARP_RMS_ATTN), worst per-pixel 2.0e-3.  The gate, for EVERY tensor: rms <= 2e-4 (0.12 .. 0.17 x
  the fp16 floor; attention outputs 5e-4), pp_rel <= 2.5e-3 -- the round-4 verdict asked for 0.35 x floor and 1.5e-3; the per-pixel figure is an
  extreme value over up to 10^7 elements x 903 tensors, and two flips meeting in one element (2 ulps = 2.0e-3) do occur.  A wrong constant or wire in
  any layer of the executor is then visible at 1e-4 instead of hiding under the floor gate's ~1e-3 of slack (planted-bug tests below).

* WHOLE NETWORK, free-running (reported; gated at 1.0 x floor).  An fp16 pipeline of this depth is a chaotic map at the rounding level: a
  perturbation of 1e-7 (fp32 vs fp64 accumulation, nothing else changed) decorrelates the rounding realisation completely within a few layers,
  and two runs of the SAME rounding oracle then differ by 0.75..0.8 x floor (tests/test_oracle_fp16sites.py shows exactly that on the CPU).  So
  "whole-network rms <= 0.35 x floor" cannot be met by ANY two implementations that do not share their summation order bit for bit; the native
  path lands where the oracle lands against itself.  That is why the gate is layer-wise.
"""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402,F401
from forge_amd import synth  # noqa: E402

from conftest import GOLDEN, load_golden  # noqa: E402
import parity  # noqa: E402

DEV = "cuda"
SHARP_RMS = 2.0e-4           # per layer, rms(native - oracle) / rms(oracle)
SHARP_RMS_ATTN = 5.0e-4      # ... of an attention OUTPUT: the online-softmax kernels round P = exp2(s - m_running) where the oracle rounds exp2(s - max);
                             # m_running differs from the row maximum by a real number (the deferred-rescale rule, csrc/fmx_attention.hip), so the two
                             # P's are rounded at scales that are not a power of two apart -- two independent realisations of the P rounding.  Measured
                             # 1.5e-4 .. 3.4e-4 on peaked softmaxes (text context, small latents); the one-pass short-context kernel, which uses the row
                             # maximum itself, agrees to 2.3e-5 (tools/attn_site_probe.py).  Still <= 0.35 x the fp16 floor.
SHARP_PP = 2.5e-3            # per layer, max |d| / max(|oracle|, rms(oracle)): 2.5 fp16 ulps of a value at the top of its binade
SHARP_RMS_FACTOR = 0.35      # the verdict's whole-network figure, kept for the planted-bug arithmetic: 0.35 x the fp16 floor's rms
WHOLE_NET_FACTOR = 1.0       # free-running whole network: two realisations of one rounding process (see the module docstring)

# one wrong constant each, planted in the ORACLE (the native path is right; the test has to FAIL against these)
PLANTS = {
    "GroupNorm eps 1e-6 instead of 1e-5 in one ResBlock norm (unet.py:395 vs :292)": {"gn_eps": ("input_blocks.1.0", "in_layers.0", 1e-6)},
    "tanh GELU instead of erf in one GEGLU (unet.py:111)": {"gelu_tanh": "input_blocks.1.1.transformer_blocks.0"},
}


def planted_layer(plant):
    """the tap at which a plant must show: the first stored tensor behind the wrong constant"""
    if "gn_eps" in plant:
        key, which, _ = plant["gn_eps"]
        return key + ".h" if which == "in_layers.0" else key
    return plant["gelu_tanh"] + ".ff.g"


def small_variance_state_dict(cfg, scale=0.1):
    """the tiny network with conv_in scaled down so that the first ResBlock's GroupNorm sees a variance (~3e-3) at which eps matters (the planted layer moves by ~1.4e-3: the size of the floor gate's slack): the
    only regime in which 1e-5 vs 1e-6 is more than an fp32 rounding error (rstd changes by 0.5 (1e-5 - 1e-6) / var)"""
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    sd = dict(sd)
    sd["input_blocks.0.0.weight"] = sd["input_blocks.0.0.weight"] * scale
    sd["input_blocks.0.0.bias"] = sd["input_blocks.0.0.bias"] * scale
    return sd


def _log(rec):
    path = os.environ.get("FMX_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


UP2X = set()      # the Upsample layers the LAST native forward ran as four phase convolutions (IntegratedUNet2DConditionModel.up2x_trace): a weight-rounding
                  # site the oracle has to be told about, like `fold`


def native_forward_with_taps(net, x, t, ctx, y):
    taps = {}
    net.tap = lambda name, tens: taps.__setitem__(name, tens.detach().to("cpu", copy=True))
    try:
        eps = net.forward(x.to(DEV), t.to(DEV), context=ctx.to(DEV), y=None if y is None else y.to(DEV))
    finally:
        net.tap = None
    UP2X.clear()
    UP2X.update(net.up2x_trace)
    return eps.float().cpu(), taps, dict(net.fold_trace)


def layerwise(sd, cfg, x, t, ctx, y, taps, fold, plant=None):
    """-> {layer: metrics of native vs oracle on the native layer's own inputs}"""
    from oracle import unet_fp16sites as o16
    outs, nat = {}, {}
    o16.unet_forward(sd, cfg, x, t, ctx, y, fold=fold, plant=plant, teacher=taps, layer_out=outs, native_view=nat, up2x=UP2X)
    res = {key: parity.metrics(nat[key], ref) for key, ref in outs.items() if key in nat}
    missing = set(taps) - set(outs)
    assert not missing, f"executor layers the oracle did not visit: {sorted(missing)}"
    return res


def _kind(key):
    for suffix, kind in ((".attn1.q", "q / k / v projections"), (".attn1.k", "q / k / v projections"), (".attn1.v", "q / k / v projections"),
                         (".attn2.q", "q / k / v projections"), (".attn2.k", "q / k / v projections"), (".attn2.v", "q / k / v projections"),
                         (".attn1.o", "self-attention output"), (".attn2.o", "cross-attention output"), (".attn1", "stream after attn1"), (".attn2", "stream after attn2"),
                         (".ff.g", "GEGLU output"), (".proj_in", "proj_in"), (".h", "ResBlock conv1 + emb")):
        if key.endswith(suffix):
            return kind
    if ".transformer_blocks." in key:
        return "stream after the feed-forward"
    return {"time_embed": "time / label embedding", "out.2": "output head", "input_blocks.0.0": "conv_in"}.get(key, "ResBlock / SpatialTransformer / Down / Up output")


def by_kind(res):
    """-> {layer kind: {layers, worst rms_rel, worst pp_rel}}"""
    out = {}
    for k, m in res.items():
        d = out.setdefault(_kind(k), {"layers": 0, "worst_rms_rel": 0.0, "worst_pp_rel": 0.0})
        d["layers"] += 1
        d["worst_rms_rel"] = max(d["worst_rms_rel"], round(m["rms_rel"], 8))
        d["worst_pp_rel"] = max(d["worst_pp_rel"], round(m["pp_rel"], 8))
    return out


def gates(key):
    return (SHARP_RMS_ATTN if key.endswith((".attn1.o", ".attn2.o")) else SHARP_RMS), SHARP_PP


def over_gate(res):
    return sorted(k for k, v in res.items() if v["rms_rel"] > gates(k)[0] or v["pp_rel"] > gates(k)[1])


def _worst(res):
    kr = max(res, key=lambda k: res[k]["rms_rel"])
    kp = max(res, key=lambda k: res[k]["pp_rel"])
    return kr, res[kr]["rms_rel"], kp, res[kp]["pp_rel"]


def _run_case(name, cfg, sd, x, t, ctx, y, floor_key, whole_network=True):
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    from oracle import unet_fp16sites as o16
    net = IntegratedUNet2DConditionModel(cfg, sd, device=DEV)
    eps, taps, fold = native_forward_with_taps(net, x, t, ctx, y)
    t0 = time.time()
    res = layerwise(sd, cfg, x, t, ctx, y, taps, fold)
    kr, rms, kp, pp = _worst(res)
    nf = sum(int(f) for v in fold.values() for f in v)
    fl = parity.FLOORS[floor_key]
    rec = {"name": f"sharp layer-wise: {name}", "layers": len(res), "folded_norms": nf, "upsample_convs_as_phase_convolutions": sorted(UP2X), "worst_rms_rel": round(rms, 8), "worst_rms_layer": kr,
           "worst_pp_rel": round(pp, 8), "worst_pp_layer": kp, "median_rms_rel": round(sorted(v["rms_rel"] for v in res.values())[len(res) // 2], 8),
           "floor_rms_rel": fl["rms_rel"], "worst_rms_over_floor": round(rms / fl["rms_rel"], 4), "gate": {"rms_rel": SHARP_RMS, "rms_rel_attention_outputs": SHARP_RMS_ATTN, "pp_rel": SHARP_PP},
           "oracle_seconds": round(time.time() - t0, 1), "by_kind": by_kind(res)}
    if whole_network:
        t0 = time.time()
        free = o16.unet_forward(sd, cfg, x, t, ctx, y, fold=fold, up2x=UP2X)
        m = parity.metrics(eps, free)
        rec["whole_network_free_running"] = {**{k: round(v, 7) for k, v in m.items()}, "rms_over_floor": round(m["rms_rel"] / fl["rms_rel"], 4),
                                            "oracle_seconds": round(time.time() - t0, 1)}
    print("[sharp]", json.dumps(rec))
    _log(rec)
    bad = over_gate(res)
    assert not bad, (name, {k: res[k] for k in bad[:5]})
    if whole_network:
        assert rec["whole_network_free_running"]["rms_rel"] <= WHOLE_NET_FACTOR * fl["rms_rel"], rec
    return net, taps, fold, res


@pytest.mark.parametrize("name", ["tiny_sd15", "tiny_sdxl"])
def test_tiny_networks_layer_by_layer(name):
    cfg = {"tiny_sd15": synth.TINY_SD15_UNET_CONFIG, "tiny_sdxl": synth.TINY_SDXL_UNET_CONFIG}[name]
    g = load_golden(f"{name}_unet_fwd.pt")
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    _run_case(name, cfg, sd, g["x"], g["t"], g["ctx"], g["y"], f"{name}_unet_fwd.pt:eps")


def test_sd15_full_size_forward_layer_by_layer():
    """BASELINE config 0 / 2's network at full size (860 M parameters, 64x64 latent), batch 2.  At this batch the projections that write the
    stream run on 4-wave tiles that emit no row statistics, so every LayerNorm runs as its own kernel (`folded_norms` = 0 in the record); the
    folds are exercised by test_sdxl_bench_batch_with_the_layernorm_folds_live_layer_by_layer below."""
    from oracle.make_golden import _inputs
    cfg = synth.SD15_UNET_CONFIG
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    x, t, ctx, y = _inputs(cfg, 2, 64, seed=11)
    _run_case("SD1.5 full size, batch 2", cfg, sd, x, t, ctx, y, "sd15_config0.pt:latent", whole_network=False)


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "sdxl_full_fwd.pt")), reason="full fixture not generated")
def test_sdxl_full_size_forward_layer_by_layer(sdxl_sd):
    """The bench workload's network (2.57 B parameters, 128x128 latent), the reference fixture's own input: every one of its 70 transformer blocks,
    17 ResBlocks ... against the rounding oracle, layer by layer, on the executor's own inputs."""
    from oracle.make_golden import _inputs
    g = load_golden("sdxl_full_fwd.pt")
    cfg = synth.SDXL_UNET_CONFIG
    x, t, ctx, y = _inputs(cfg, 1, 128, seed=g["inputs_seed"])
    _run_case("SDXL full size (128x128 latent)", cfg, sdxl_sd, x, t, ctx, y, "sdxl_full_fwd.pt:eps", whole_network=False)


# ---- the configuration the bench is timed on: UNet batch 16 on the 128 x 128 latent, every LayerNorm folded ---------------------------
BENCH_BATCH = 16
BENCH_IMAGES = (5, 14)      # the images the oracle replays (teacher forcing makes images independent: every layer is evaluated on the executor's own
                            # stored input of THAT image; LayerNorm / GroupNorm statistics are per row / per image)
FOLD_PLANT_BLOCK = "input_blocks.4.1.transformer_blocks.0"     # 640 channels, 64 x 64 tokens: M = 65 536 rows at batch 16


def fold_plants(block=FOLD_PLANT_BLOCK):
    return {
        "LayerNorm eps 1e-6 instead of 1e-5 inside one folded norm2 (unet.py:167-175; nn.LayerNorm's default)": {"ln_eps": (block, "norm2", 1e-6)},
        "colsum(W) instead of colsum(W * gamma) in one folded ff.net.0 (the mean term of the fold)": {"colsum_unscaled": (block, "norm3")},
    }


def fold_planted_layer(plant):
    """the first tensor stored behind the wrong constant"""
    return plant["ln_eps"][0] + ".attn2.q" if "ln_eps" in plant else plant["colsum_unscaled"][0] + ".ff.g"


def bench_batch_inputs(cfg, b=BENCH_BATCH, hw=128, seed=29):
    """sixteen DIFFERENT images, conditionings and timesteps (a repeated input would let a batch-position bug through)"""
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(b, cfg["in_channels"], hw, hw, generator=g)
    ctx = torch.randn(b, 77, cfg["context_dim"], generator=g)
    y = torch.randn(b, cfg["adm_in_channels"], generator=g)
    t = torch.linspace(991.0, 7.0, b).round()
    return x, t, ctx, y


def small_stream_state_dict(sd, block=FOLD_PLANT_BLOCK, scale=0.04):
    """the stream in front of `block`'s norm2 scaled down (proj_in and attn1.to_out x scale: variance ~2e-3), the only regime in which a LayerNorm
    eps of 1e-6 vs 1e-5 is more than an fp32 rounding error (rstd moves by 0.5 * 9e-6 / var).  A shallow copy: four tensors replaced."""
    st_key = block.split(".transformer_blocks.")[0]
    out = dict(sd)
    for k in (st_key + ".proj_in", block + ".attn1.to_out.0"):
        out[k + ".weight"] = sd[k + ".weight"] * scale
        out[k + ".bias"] = sd[k + ".bias"] * scale
    return out


@pytest.fixture(scope="module")
def sdxl_sd():
    return synth.synth_unet_state_dict(synth.SDXL_UNET_CONFIG, seed=0)


def native_forward_with_taps_of(net, x, t, ctx, y, images, prefix=""):
    """the forward at the full batch; of every stored tensor only `images` (and only layers under `prefix`) leave the device"""
    from forge_amd import hipops
    taps = {}
    sel = torch.tensor(images, device=DEV)

    def tap(name, tens):
        if name.startswith(prefix) or name == "time_embed":
            taps[name] = tens.index_select(0, sel).to("cpu")
    net.tap = tap
    before = hipops.LN_FOLDED_LAUNCHES
    try:
        eps = net.forward(x.to(DEV), t.to(DEV), context=ctx.to(DEV), y=y.to(DEV))
    finally:
        net.tap = None
    UP2X.clear()
    UP2X.update(net.up2x_trace)
    return eps.float().cpu(), taps, dict(net.fold_trace), hipops.LN_FOLDED_LAUNCHES - before


def test_sdxl_bench_batch_with_the_layernorm_folds_live_layer_by_layer(sdxl_sd):
    """What bench.py times: the SDXL network at UNet batch 16 on the 128 x 128 latent -- 256 x 320 tiles everywhere, so norm1 / norm2 / norm3 of all
    70 transformer blocks run FOLDED into the projections around them (`gemm256p_kernel<..., LN = 1 | 2 | 3>`: 48.6 % of the bench's GPU time), and the
    GroupNorm statistics come from the GEMM epilogues.  The fold computes rstd (x W'^T - mean colsum(W')) + bias' -- a cancellation -- from row sums
    the producing GEMM left; the oracle evaluates the same formula at the same rounding sites (oracle/unet_fp16sites.py _folded_linear) on the
    executor's stored stream, layer by layer, for two of the sixteen images."""
    from forge_amd.backend.nn import unet as _unet
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    cfg = synth.SDXL_UNET_CONFIG
    x, t, ctx, y = bench_batch_inputs(cfg)
    net = IntegratedUNet2DConditionModel(cfg, sdxl_sd, device=DEV)
    eps, taps, fold, launches = native_forward_with_taps_of(net, x, t, ctx, y, BENCH_IMAGES)
    del net
    torch.cuda.empty_cache()
    nf = sum(int(f) for v in fold.values() for f in v)
    if _unet._LN_FOLD and _unet._LN_FOLD1:
        assert nf == 3 * 70 and launches == 4 * 70, (nf, launches)     # q|k, V^T, attn2.to_q, ff.net.0 per block
    sel = list(BENCH_IMAGES)
    t0 = time.time()
    res = layerwise(sdxl_sd, cfg, x[sel], t[sel], ctx[sel], y[sel], taps, fold)
    kr, rms, kp, pp = _worst(res)
    fl = parity.FLOORS["sdxl_full_fwd.pt:eps"]
    rec = {"name": f"sharp layer-wise: SDXL full size at the bench's UNet batch {BENCH_BATCH}, images {sel} (LayerNorm folds live)", "layers": len(res),
           "folded_norms": nf, "folded_gemm_launches": launches, "upsample_convs_as_phase_convolutions": sorted(UP2X), "worst_rms_rel": round(rms, 8), "worst_rms_layer": kr, "worst_pp_rel": round(pp, 8),
           "worst_pp_layer": kp, "median_rms_rel": round(sorted(v["rms_rel"] for v in res.values())[len(res) // 2], 8), "floor_rms_rel": fl["rms_rel"],
           "worst_rms_over_floor": round(rms / fl["rms_rel"], 4), "gate": {"rms_rel": SHARP_RMS, "rms_rel_attention_outputs": SHARP_RMS_ATTN, "pp_rel": SHARP_PP},
           "oracle_seconds": round(time.time() - t0, 1), "by_kind": by_kind(res)}
    print("[sharp]", json.dumps(rec))
    _log(rec)
    assert nf > 0, "the folds are what this case exists for"
    bad = over_gate(res)
    assert not bad, {k: res[k] for k in bad[:5]}


def test_planted_bugs_inside_the_layernorm_fold_fail_at_their_layer(sdxl_sd):
    """Two wrong constants INSIDE the fold, planted in the oracle's copy of ONE block at the bench's sizes: a LayerNorm eps of 1e-6 in the folded norm2
    (visible because that block's stream is scaled down to a variance where eps matters) and the mean term's column sums taken over W instead of
    W * gamma in the folded ff.net.0.  The native executor must agree with the unplanted oracle on every layer of the block and disagree with each
    planted one at exactly the first tensor stored behind the wrong constant."""
    from oracle import unet_fp16sites as o16
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    cfg = synth.SDXL_UNET_CONFIG
    sd = small_stream_state_dict(sdxl_sd)
    x, t, ctx, y = bench_batch_inputs(cfg)
    net = IntegratedUNet2DConditionModel(cfg, sd, device=DEV)
    st_key = FOLD_PLANT_BLOCK.split(".transformer_blocks.")[0]
    _, taps, fold, _ = native_forward_with_taps_of(net, x, t, ctx, y, BENCH_IMAGES, prefix=st_key)
    del net
    torch.cuda.empty_cache()
    assert fold[FOLD_PLANT_BLOCK] == (True, True, True), fold[FOLD_PLANT_BLOCK]
    sel = list(BENCH_IMAGES)

    def block_metrics(plant):
        outs, nat = {}, {}
        o16.transformer_block_only(sd, cfg, FOLD_PLANT_BLOCK, ctx[sel], fold, taps, plant=plant, layer_out=outs, native_view=nat)
        return {k: parity.metrics(nat[k], ref) for k, ref in outs.items() if k in nat}
    good = block_metrics(None)
    assert len(good) == 12 and not over_gate(good), good
    var = float(taps[FOLD_PLANT_BLOCK + ".attn1"].float().var(dim=-1).mean())
    for pname, plant in fold_plants().items():
        bad = block_metrics(plant)
        layer = fold_planted_layer(plant)
        failing = over_gate(bad)
        rec = {"name": f"planted bug in the fold: {pname}", "layer": layer, "sharp_gate_fails_at": failing, "planted_layer_rms_rel": round(bad[layer]["rms_rel"], 7),
               "planted_layer_pp_rel": round(bad[layer]["pp_rel"], 7), "unplanted_layer_rms_rel": round(good[layer]["rms_rel"], 8),
               "stream_variance_in_front_of_norm2": round(var, 6), "unet_batch": BENCH_BATCH, "images": sel}
        print("[sharp]", json.dumps(rec))
        _log(rec)
        assert failing == [layer], rec


@pytest.mark.parametrize("pname", list(PLANTS))
def test_planted_bug_fails_the_sharp_gate_and_passes_the_floor_gate(pname):
    """A wrong constant planted in the ORACLE's copy of ONE layer: the layer-wise gate must fail AT that layer (and nowhere else), while the gate the
    suite used until round 4 -- whole network against the reference fixture at 1.25 x floor -- is recorded for the same wrong constant planted
    the other way round (a native path with that bug would have produced the planted oracle's output up to the sharp gate)."""
    from oracle import unet_fp16sites as o16
    from forge_amd.backend.nn.unet import IntegratedUNet2DConditionModel
    plant = PLANTS[pname]
    cfg = synth.TINY_SD15_UNET_CONFIG
    g = load_golden("tiny_sd15_unet_fwd.pt")
    sd = small_variance_state_dict(cfg) if "gn_eps" in plant else synth.synth_unet_state_dict(cfg, seed=0)
    net = IntegratedUNet2DConditionModel(cfg, sd, device=DEV)
    eps, taps, fold = native_forward_with_taps(net, g["x"], g["t"], g["ctx"], g["y"])
    good = layerwise(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], taps, fold)
    bad = layerwise(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], taps, fold, plant=plant)
    layer = planted_layer(plant)
    assert not over_gate(good)
    failing = over_gate(bad)
    # would the floor gate have seen it?  the planted network, free-running, against the unplanted fp32 restatement, held to what check() allows
    from oracle import unet as ou
    ref32 = ou.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"])
    planted_free = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], fold=fold, plant=plant)
    unplanted_free = o16.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"], g["y"], fold=fold)
    mp, mu = parity.metrics(planted_free, ref32), parity.metrics(unplanted_free, ref32)
    _, lim = parity.limits("tiny_sd15_unet_fwd.pt:eps")
    floor_gate_catches = any(mp[k] > lim[k] for k in lim)
    rec = {"name": f"planted bug: {pname}", "layer": layer, "sharp_gate_fails_at": failing, "planted_layer_rms_rel": round(bad[layer]["rms_rel"], 7),
           "planted_layer_pp_rel": round(bad[layer]["pp_rel"], 7), "unplanted_layer_rms_rel": round(good[layer]["rms_rel"], 8),
           "floor_gate_would_catch_it": floor_gate_catches, "planted_vs_fp32": {k: round(v, 6) for k, v in mp.items()},
           "unplanted_vs_fp32": {k: round(v, 6) for k, v in mu.items()}, "floor_gate_limit": {k: round(v, 6) for k, v in lim.items()}}
    print("[sharp]", json.dumps(rec))
    _log(rec)
    assert failing == [layer], rec
