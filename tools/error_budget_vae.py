"""Per-site fp16 error budget of the SDXL VAE decode at 1024^2 (round 4, VERDICT r3 item 7: "measure an fp32-GroupNorm-input variant's pp_rel on sdxl_vae1024.pt and
its cost") -- CPU study on the pinned restatement (oracle/vae.py arithmetic, fp32) with fp16 ROUNDING inserted at selectable sites, against the committed
reference output (tests/golden/sdxl_vae1024.pt: every 4th pixel + one crop).  ANALYSIS INFRASTRUCTURE ONLY.

    w   conv weights rounded to fp16
    a   what a convolution READS (GroupNorm + SiLU output, the upsampled tensor, the attention's q / k / v / output) rounded to fp16
    s   the RESIDUAL STREAM -- every tensor a GroupNorm reads: conv_in / conv2 / upsample-conv outputs and the residual sums -- rounded to fp16
    `was` = the fp16 executor as shipped;  `wa` = "fp32 GroupNorm inputs" (the stream and the conv outputs that feed the norms kept in fp32)

usage: python tools/error_budget_vae.py [policy ...]      default: was wa s a w -
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
from oracle.attention import attention_single_head_spatial  # noqa: E402
import parity  # noqa: E402


def r16(x):
    return x.half().float()


def decode(sd, z, pol):
    W = (lambda k: r16(sd[k])) if "w" in pol else (lambda k: sd[k])
    A = r16 if "a" in pol else (lambda x: x)
    S = r16 if "s" in pol else (lambda x: x)

    def gn(key, x):
        return F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], 1e-6)

    def conv(key, x, padding=1):
        return F.conv2d(x, W(key + ".weight"), sd[key + ".bias"], padding=padding)

    def res(key, x):
        h = S(conv(key + ".conv1", A(F.silu(gn(key + ".norm1", x)))))        # conv1's output is read by norm2
        h = conv(key + ".conv2", A(F.silu(gn(key + ".norm2", h))))
        if key + ".nin_shortcut.weight" in sd:
            x = conv(key + ".nin_shortcut", A(x), padding=0)
        return S(x + h)

    def attn(key, x):
        h = A(gn(key + ".norm", x))
        q, k, v = A(conv(key + ".q", h, 0)), A(conv(key + ".k", h, 0)), A(conv(key + ".v", h, 0))
        return S(x + conv(key + ".proj_out", A(attention_single_head_spatial(q, k, v)), 0))

    h = z
    if "post_quant_conv.weight" in sd:
        h = A(conv("post_quant_conv", A(h), 0))
    h = S(conv("decoder.conv_in", h))
    h = res("decoder.mid.block_1", h)
    h = attn("decoder.mid.attn_1", h)
    h = res("decoder.mid.block_2", h)
    nlev = 0
    while f"decoder.up.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lev in reversed(range(nlev)):
        i = 0
        while f"decoder.up.{lev}.block.{i}.norm1.weight" in sd:
            h = res(f"decoder.up.{lev}.block.{i}", h)
            i += 1
        if lev != 0:
            h = S(conv(f"decoder.up.{lev}.upsample.conv", A(F.interpolate(h, scale_factor=2.0, mode="nearest"))))
    return r16(conv("decoder.conv_out", A(F.silu(gn("decoder.norm_out", h)))))


def main():
    pols = sys.argv[1:] or ["was", "wa", "s", "a", "w", "-"]
    vcfg = synth.SDXL_VAE_CONFIG
    sd = synth.synth_vae_decoder_state_dict(vcfg, seed=1)
    g = torch.load(os.path.join(ROOT, "tests", "golden", "sdxl_vae1024.pt"), map_location="cpu", weights_only=False)
    z = g["latent"] / vcfg["scaling_factor"] + vcfg.get("shift_factor", 0.0) if False else None
    from oracle.vae import process_out
    z = process_out(g["latent"], vcfg["scaling_factor"], vcfg.get("shift_factor") or 0.0)
    want = torch.cat([g["decoded_s4"].reshape(-1), g["decoded_crop"].reshape(-1)])
    for pol in pols:
        t0 = time.time()
        with torch.no_grad():
            d = decode(sd, z, pol)
        got = torch.cat([d[:, :, ::4, ::4].reshape(-1), d[:, :, 448:576, 448:576].reshape(-1)])
        m = parity.metrics(got, want)
        m.update(parity.unclamped(got, want))
        print(json.dumps({"vae": "SDXL decoder, 1024x1024 (tests/golden/sdxl_vae1024.pt)", "fp16_rounding_sites": pol, **{k: round(v, 6) for k, v in m.items()},
                          "cpu_seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
