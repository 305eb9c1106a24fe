"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates backend/attention.py:37-93 (`attention_basic`, the explicit softmax(QK^T/sqrt(d))V form that
`attention_pytorch` :324-339 is numerically equivalent to) and the VAE's single-head spatial variant
(:342-427, scale C^-0.5).  fp32 throughout; no mask on the UNet path.
"""
import torch


def attention(q, k, v, heads):
    """q [B,Nq,h*d], k,v [B,Nk,h*d] -> [B,Nq,h*d]"""
    b, nq, c = q.shape
    d = c // heads
    q = q.reshape(b, nq, heads, d).permute(0, 2, 1, 3)
    k = k.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    v = v.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    p = sim.softmax(dim=-1)
    out = torch.matmul(p, v)
    return out.permute(0, 2, 1, 3).reshape(b, nq, c)


def attention_single_head_spatial(q, k, v):
    """q,k,v [B,C,H,W] -> [B,C,H,W]; one head of width C (attention.py:412-422)."""
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).transpose(1, 2)
    k = k.reshape(b, c, h * w).transpose(1, 2)
    v = v.reshape(b, c, h * w).transpose(1, 2)
    out = attention(q, k, v, 1)
    return out.transpose(1, 2).reshape(b, c, h, w)
