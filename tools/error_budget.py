"""Per-site fp16 error budget of the UNet forward (round 3) -- CPU study, TEST/ANALYSIS INFRASTRUCTURE ONLY.

Question: which rounding sites of an fp16 executor carry the 3-6e-3 per-pixel error against the reference's fp32 run, and what does an
executor have to keep in fp32 to reach the north star's 1e-3 per pixel?  This tool re-runs the pinned CPU restatement (oracle/unet.py, fp32
arithmetic) with fp16 ROUNDING inserted at selectable sites and compares each variant with the committed reference output:

    w   conv / linear WEIGHTS rounded to fp16                  (the executor stores fp16 weights)
    a   GEMM A-operands (the activation a conv / linear reads) rounded to fp16
    o   branch tensors rounded where they are written: q, k, v, the attention output, GEGLU output, conv1's output (what GroupNorm 2 reads)
    s   the RESIDUAL STREAM rounded after every residual add   (x = fp16(x + f(x)): ResBlock, the three transformer sub-layers, proj_out)
    p   softmax probabilities rounded to fp16 before P.V

`wa os p` all on = the executor as shipped in round 2 (fp32 accumulation everywhere);  `s` off = "precise" mode (fp32 residual stream);
`w` off additionally = what a hi/lo-split weight operand would give.

usage: python tools/error_budget.py sd15|sdxl|tiny_sd15|tiny_sdxl [policy ...]     (policy = subset string of "waosp", "-" = none)
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import forge_amd  # noqa: F401,E402
from forge_amd import synth  # noqa: E402
from oracle import unet as ou  # noqa: E402
from oracle.make_golden import _inputs  # noqa: E402
import parity  # noqa: E402


def r16(x):
    return x.half().float()


class Policy:
    def __init__(self, s):
        self.w, self.a, self.o, self.s, self.p = ("w" in s), ("a" in s), ("o" in s), ("s" in s), ("p" in s)
        self.name = s
        self._wcache = {}

    def W(self, sd, key):
        if not self.w:
            return sd[key]
        if key not in self._wcache:
            self._wcache[key] = r16(sd[key])
        return self._wcache[key]

    def A(self, x):
        return r16(x) if self.a else x

    def O(self, x):
        return r16(x) if self.o else x

    def S(self, x):
        return r16(x) if self.s else x


POL = Policy("")


def conv(sd, key, x, stride=1, padding=1):
    return F.conv2d(POL.A(x), POL.W(sd, key + ".weight"), sd[key + ".bias"], stride=stride, padding=padding)


def lin(sd, key, x):
    return F.linear(POL.A(x), POL.W(sd, key + ".weight"), sd.get(key + ".bias"))


def attention(q, k, v, heads):
    b, nq, c = q.shape
    d = c // heads
    q, k, v = (POL.O(t) for t in (q, k, v))
    q = q.reshape(b, nq, heads, d).permute(0, 2, 1, 3)
    k = k.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    v = v.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    m = sim.amax(dim=-1, keepdim=True)
    e = torch.exp(sim - m)
    l = e.sum(dim=-1, keepdim=True)
    if POL.p:
        e = r16(e)
    out = torch.matmul(e, v) / l
    return POL.O(out.permute(0, 2, 1, 3).reshape(b, nq, c))


def resblock(sd, key, x, emb):
    h = conv(sd, key + ".in_layers.2", F.silu(ou._gn(sd, key + ".in_layers.0", x, 1e-5)))
    e = ou._lin(sd, key + ".emb_layers.1", F.silu(emb))
    h = POL.O(h + e[:, :, None, None])
    h = conv(sd, key + ".out_layers.3", F.silu(ou._gn(sd, key + ".out_layers.0", h, 1e-5)))
    if key + ".skip_connection.weight" in sd:
        x = conv(sd, key + ".skip_connection", x, padding=0)
    return POL.S(x + h)


def cross_attention(sd, key, x, context, heads):
    q = lin(sd, key + ".to_q", x)
    ctx = x if context is None else context
    k = lin(sd, key + ".to_k", ctx)
    v = lin(sd, key + ".to_v", ctx)
    return lin(sd, key + ".to_out.0", attention(q, k, v, heads))


def transformer_block(sd, key, x, context, heads, to=None):
    x = POL.S(x + cross_attention(sd, key + ".attn1", ou._ln(sd, key + ".norm1", x), None, heads))
    x = POL.S(x + cross_attention(sd, key + ".attn2", ou._ln(sd, key + ".norm2", x), context, heads))
    h = lin(sd, key + ".ff.net.0.proj", ou._ln(sd, key + ".norm3", x))
    a, gate = h.chunk(2, dim=-1)
    h = POL.O(a * F.gelu(gate))
    return POL.S(x + lin(sd, key + ".ff.net.2", h))


def spatial_transformer(sd, key, x, context, heads, to=None):
    b, c, hh, ww = x.shape
    x_in = x
    x = ou._gn(sd, key + ".norm", x, 1e-6)
    use_linear = sd[key + ".proj_in.weight"].ndim == 2
    if not use_linear:
        x = conv(sd, key + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
    if use_linear:
        x = lin(sd, key + ".proj_in", x)
    x = POL.S(x)
    d = 0
    while f"{key}.transformer_blocks.{d}.norm1.weight" in sd:
        x = transformer_block(sd, f"{key}.transformer_blocks.{d}", x, context, heads, to)
        d += 1
    if use_linear:
        x = lin(sd, key + ".proj_out", x)
    x = x.reshape(b, hh, ww, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = conv(sd, key + ".proj_out", x, padding=0)
    return POL.S(x + x_in)


def install():
    """route the oracle's block functions through the policy-aware ones (the structure walk, skip concatenation, timestep / label MLPs stay
    the oracle's own; plain convs -- conv_in, down / upsample, out.2 -- get operand rounding through `_conv`)"""
    ou.resblock, ou.transformer_block, ou.spatial_transformer = resblock, transformer_block, spatial_transformer
    ou._conv = lambda sd, key, x, stride=1, padding=1: POL.S(conv(sd, key, x, stride, padding)) if key != "out.2" else conv(sd, key, x, stride, padding)


CASES = {
    "sd15": (synth.SD15_UNET_CONFIG, "sd15_config0.pt", None),
    "sdxl": (synth.SDXL_UNET_CONFIG, "sdxl_full_fwd.pt", 128),
    "tiny_sd15": (synth.TINY_SD15_UNET_CONFIG, "tiny_sd15_unet_fwd.pt", None),
    "tiny_sdxl": (synth.TINY_SDXL_UNET_CONFIG, "tiny_sdxl_unet_fwd.pt", None),
}


def main():
    global POL
    case = sys.argv[1]
    pols = sys.argv[2:] or ["-", "waosp", "waop", "aop", "wa", "w", "a", "s", "o", "p"]
    cfg, fixture, hw = CASES[case]
    g = torch.load(os.path.join(ROOT, "tests", "golden", fixture), weights_only=False)
    if "x" in g:
        x, t, ctx, y = g["x"], g["t"], g["ctx"], g.get("y")
    else:
        x, t, ctx, y = _inputs(cfg, 1, hw, seed=g["inputs_seed"])
    sd = synth.synth_unet_state_dict(cfg, seed=0)
    install()
    out_path = os.environ.get("ERROR_BUDGET_OUT")
    for s in pols:
        POL = Policy("" if s == "-" else s)
        t0 = time.time()
        with torch.no_grad():
            eps = ou.unet_forward(sd, cfg, x, t, ctx, y)
        m = parity.metrics(eps, g["eps"])
        rec = {"case": case, "policy": s, **{k: float(f"{v:.4g}") for k, v in m.items()}, "seconds": round(time.time() - t0, 1)}
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
