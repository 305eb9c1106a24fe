"""TEST INFRASTRUCTURE ONLY (CPU oracle): the T5 encoder with rounding at the native executor's storage sites, in the executor's element type.

The companion of oracle/unet_fp16sites.py for the T5-XXL text encoder of Flux (reference backend/nn/t5.py:15-214).  The arithmetic is oracle/t5.py's
(pinned to the reference's module through tests/golden/tiny_t5.pt); `dtype=None` reproduces it bit for bit.  With `dtype` = torch.bfloat16 / torch.float16
what the native executor (stable-diffusion-webui-forge_amd/backend/nn/t5.py) stores is rounded where it stores it: parameters (the relative-position bias table
stays fp32, its gathered [H, T, T] bias is rounded), the gathered token embeddings, both RMS norms, q|k, V^T, the attention output (generic masked kernel:
fp32 scores = q k^T + bias, no scale; P rounded for P V; row sum over the unrounded exponentials), o W_o + x with one rounding, wi_1 x, gelu_tanh(wi_0 x) x
(wi_1 x) with one rounding (the gate lives in the GEMM epilogue), h W_o + x with one rounding, the final norm.
`teacher` = the native executor's block outputs: block i is evaluated on the NATIVE output of block i - 1 (layer-wise comparison)."""
import torch
import torch.nn.functional as F

from . import t5 as ot


@torch.no_grad()
def t5_block_outputs(sd, cfg, ids, dtype=torch.bfloat16, teacher=None):
    """-> [embeddings, after block 1, ..., after block N, final norm output]"""
    R = (lambda t: t.to(dtype).float()) if dtype is not None else (lambda t: t)
    p = "transformer.encoder.block."
    H = cfg["num_heads"]
    b, t = ids.shape
    x = R(sd["transformer.shared.weight"])[ids]
    c = x.shape[-1]
    d = c // H
    bias = R(ot.position_bias(sd[p + "0.layer.0.SelfAttention.relative_attention_bias.weight"].float(), t))
    outs = [x]
    for i in range(cfg["num_layers"]):
        if teacher is not None:
            x = teacher[i].float().reshape(b, t, c)
        a = f"{p}{i}.layer.0."
        n = R(ot.rms_norm(x, R(sd[a + "layer_norm.weight"])))
        q, k, v = (R(F.linear(n, R(sd[a + f"SelfAttention.{w}.weight"]))) for w in ("q", "k", "v"))
        if dtype is None:
            o = ot._attn_with_bias(q, k, v, H, bias)
        else:
            qh, kh, vh = (z.reshape(b, t, H, d).permute(0, 2, 1, 3) for z in (q, k, v))
            s = (torch.matmul(qh, kh.transpose(-1, -2)) + bias) * 1.4426950408889634
            pr = torch.exp2(s - s.amax(dim=-1, keepdim=True))
            o = R((torch.matmul(R(pr), vh) / pr.sum(dim=-1, keepdim=True)).permute(0, 2, 1, 3).reshape(b, t, c))
        x = R(x + F.linear(o, R(sd[a + "SelfAttention.o.weight"])))
        f = f"{p}{i}.layer.1."
        n = R(ot.rms_norm(x, R(sd[f + "layer_norm.weight"])))
        lin = R(F.linear(n, R(sd[f + "DenseReluDense.wi_1.weight"])))
        h = R(F.gelu(F.linear(n, R(sd[f + "DenseReluDense.wi_0.weight"])), approximate="tanh") * lin)
        x = R(x + F.linear(h, R(sd[f + "DenseReluDense.wo.weight"])))
        outs.append(x)
    if teacher is not None:
        x = teacher[cfg["num_layers"]].float().reshape(b, t, c)
    outs.append(R(ot.rms_norm(x, R(sd["transformer.encoder.final_layer_norm.weight"]))))
    return outs
