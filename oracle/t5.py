"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- never imported by the product package.

Torch-fp32 restatement of the reference's T5 encoder, backend/nn/t5.py:15-214 (T5LayerNorm, gated tanh-GELU feed-forward, self-attention with the
bucketed relative-position bias of block 0 reused by every block, no attention scaling), as plain functions over an LDM / transformers-keyed state dict.
Pinned against the imported reference class on the tiny configuration (tests/golden/tiny_t5.pt, oracle/make_golden.py gen_t5; tests/test_oracle_golden.py)."""
import math

import torch
import torch.nn.functional as F

from .attention import attention


def rms_norm(x, w, eps=1e-6):
    # t5.py:21-25
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    # t5.py:88-110, bidirectional
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def position_bias(table, t):
    """table [32, H] -> additive bias [1, H, t, t] (t5.py:112-125)"""
    pos = torch.arange(t)
    buckets = relative_position_bucket(pos[None, :] - pos[:, None])
    return table[buckets].permute(2, 0, 1).unsqueeze(0)


def _attn_with_bias(q, k, v, heads, bias):
    b, n, c = q.shape
    d = c // heads
    qh, kh, vh = (t.reshape(b, n, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    sim = torch.matmul(qh, kh.transpose(-1, -2)) + bias          # the reference multiplies k by sqrt(d) to cancel SDPA's 1/sqrt(d) (:137): no scale
    return torch.matmul(sim.softmax(-1), vh).permute(0, 2, 1, 3).reshape(b, n, c)


@torch.no_grad()
def t5_encode(sd, cfg, ids):
    """ids [B, T] -> [B, T, d_model] (T5.forward -> T5Stack, t5.py:163-209; no attention mask: the engine passes none, t5_engine.py:59-66)"""
    p = "transformer.encoder.block."
    H = cfg["num_heads"]
    x = sd["transformer.shared.weight"][ids]
    bias = None
    for i in range(cfg["num_layers"]):
        a = f"{p}{i}.layer.0."
        n = rms_norm(x, sd[a + "layer_norm.weight"])
        if a + "SelfAttention.relative_attention_bias.weight" in sd:
            bias = position_bias(sd[a + "SelfAttention.relative_attention_bias.weight"], ids.shape[1])
        q, k, v = (F.linear(n, sd[a + f"SelfAttention.{w}.weight"]) for w in ("q", "k", "v"))
        x = x + F.linear(_attn_with_bias(q, k, v, H, bias), sd[a + "SelfAttention.o.weight"])
        f = f"{p}{i}.layer.1."
        n = rms_norm(x, sd[f + "layer_norm.weight"])
        if cfg.get("is_gated_act", True):
            h = F.gelu(F.linear(n, sd[f + "DenseReluDense.wi_0.weight"]), approximate="tanh") * F.linear(n, sd[f + "DenseReluDense.wi_1.weight"])
        else:
            h = F.gelu(F.linear(n, sd[f + "DenseReluDense.wi.weight"]), approximate="tanh")
        x = x + F.linear(h, sd[f + "DenseReluDense.wo.weight"])
    return rms_norm(x, sd["transformer.encoder.final_layer_norm.weight"])
