"""Which rounding sites does an attention launch have?  Compares fmx_attention_f16 on one cross-attention and one self-attention shape with variants
of the reference arithmetic (backend/attention.py:324-339) that differ only in WHERE fp16 rounding happens -- used once to make
oracle/unet_fp16sites.py faithful to the short-context kernel (round 5).  usage: python tools/attn_site_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops

DEV = "cuda"
LOG2E = 1.44269504088896340736


def r16(x):
    return x.half().float()


def variants(q, k, v, d):
    """q [B,H,Nq,d], k/v [B,H,Nk,d] fp16-valued fp32 -> {name: O}"""
    c = torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)
    out = {}
    for name, qs, round_p, l_from_rounded in (("q' fp16, P fp16, l from unrounded P", r16(q * c), True, False),
                                               ("q' fp16, P fp16, l from ROUNDED P", r16(q * c), True, True),
                                               ("q unscaled (scale on the fp32 scores), P fp16", None, True, False),
                                               ("q' fp16, P fp32", r16(q * c), False, False),
                                               ("all fp32 (softmax(qk^T/sqrt d) v)", None, False, False)):
        s = torch.matmul(qs, k.transpose(-1, -2)) if qs is not None else torch.matmul(q, k.transpose(-1, -2)) * c
        p = torch.exp2(s - s.amax(-1, keepdim=True))
        pr = r16(p) if round_p else p
        l = (pr if l_from_rounded else p).sum(-1, keepdim=True)
        out[name] = torch.matmul(pr, v) / l
    return out


def run(b, h, nq, nk, d, dp, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(b, nq, h, d, generator=g).half()
    k = torch.randn(b, nk, h, d, generator=g).half()
    v = torch.randn(b, nk, h, d, generator=g).half()
    nkp = -(-nk // 64) * 64
    hd = h * dp
    qd = torch.zeros(b * nq, hd, dtype=torch.float16, device=DEV)
    qd.view(b, nq, h, dp)[..., :d] = q.to(DEV)
    kd = torch.zeros(b * nkp, hd, dtype=torch.float16, device=DEV)
    kd.view(b, nkp, h, dp)[:, :nk, :, :d] = k.to(DEV)
    vt = torch.zeros(hd, b * nkp, dtype=torch.float16, device=DEV)
    vt.view(h, dp, b, nkp)[:, :d, :, :nk] = v.to(DEV).permute(2, 3, 0, 1)
    o = ops.attention(qd, kd, vt, batch=b, heads=h, nq=nq, nk=nk, nk_pad=nkp, dpad=dp, scale=d ** -0.5, q_bs=nq * hd, q_rs=hd, k_bs=nkp * hd, k_rs=hd,
                      vt_bs=nkp, vt_hs=dp * b * nkp, vt_ds=b * nkp)
    torch.cuda.synchronize()
    got = o.view(b, nq, h, dp)[..., :d].float().cpu().permute(0, 2, 1, 3)
    ref = variants(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3), d)
    rec = {"shape": f"B={b} H={h} Nq={nq} Nk={nk} d={d} (padded {dp})"}
    for name, r in ref.items():
        rr = r16(r) if "all fp32" not in name else r
        rec[name] = float(((got - rr).pow(2).mean().sqrt() / rr.pow(2).mean().sqrt()))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    run(2, 8, 1024, 77, 40, 64, 1)      # SD1.5 cross-attention (heads padded 40 -> 64): the short-context kernel
    run(2, 10, 1024, 77, 64, 64, 2)     # SDXL cross-attention
    run(2, 8, 1024, 1024, 40, 64, 3)    # SD1.5 self-attention: attn_q64v2
    run(2, 4, 256, 256, 16, 64, 4)      # tiny network self-attention: attn_q64v3
    run(2, 4, 256, 77, 16, 64, 5)       # tiny network cross-attention
