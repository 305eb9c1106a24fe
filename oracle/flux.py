"""TEST INFRASTRUCTURE ONLY -- torch-fp32 CPU restatement of the reference's Flux transformer forward
(backend/nn/flux.py), functional over an LDM-style state dict.  Pinned against fixtures generated from the REAL
reference (oracle/make_golden.py gen_flux -> tests/golden/tiny_flux_fwd.pt) by tests/test_oracle_golden.py.
Never imported by the product package."""
import math

import torch
import torch.nn.functional as F


def rope(pos, dim, theta):
    # flux.py:21-40: float64 frequencies, [cos, -sin, sin, cos] 2x2 blocks, returned as fp32
    scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
    omega = 1.0 / (theta ** scale)
    out = pos.unsqueeze(-1).double() * omega.unsqueeze(0)
    out = torch.stack([torch.cos(out), -torch.sin(out), torch.sin(out), torch.cos(out)], dim=-1)
    b, n, d, _ = out.shape
    return out.view(b, n, d, 2, 2).float()


def embed_nd(ids, axes_dim, theta):
    # flux.py:76-91
    emb = torch.cat([rope(ids[..., i], axes_dim[i], theta) for i in range(ids.shape[-1])], dim=-3)
    return emb.unsqueeze(1)


def apply_rope(xq, xk, freqs_cis):
    # flux.py:43-49
    xq_ = xq.float().reshape(*xq.shape[:-1], -1, 1, 2)
    xk_ = xk.float().reshape(*xk.shape[:-1], -1, 1, 2)
    xq_out = freqs_cis[..., 0] * xq_[..., 0] + freqs_cis[..., 1] * xq_[..., 1]
    xk_out = freqs_cis[..., 0] * xk_[..., 0] + freqs_cis[..., 1] * xk_[..., 1]
    return xq_out.reshape(*xq.shape), xk_out.reshape(*xk.shape)


def timestep_embedding(t, dim, max_period=10000, time_factor=1000.0):
    # flux.py:52-73
    t = time_factor * t
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(sd, k, x):
    return F.linear(x, sd[k + ".weight"], sd.get(k + ".bias"))


def _mlp_embedder(sd, k, x):  # flux.py:94-104
    return _lin(sd, k + ".out_layer", F.silu(_lin(sd, k + ".in_layer", x)))


def _rms(x, scale, eps=1e-6):  # flux.py:107-126
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps) * scale


def _attention(q, k, v, pe):
    # flux.py:15-18 + attention_function(skip_reshape=True): softmax(q k^T / sqrt(d)) v over [B,H,L,D] -> [B,L,H*D]
    q, k = apply_rope(q, k, pe)
    d = q.shape[-1]
    s = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    o = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)
    b, h, l, _ = o.shape
    return o.permute(0, 2, 1, 3).reshape(b, l, h * d)


def _split_qkv(qkv, heads):
    b, l, _ = qkv.shape
    return qkv.view(b, l, 3, heads, -1).permute(2, 0, 3, 1, 4)


def _modulation(sd, k, vec, n):  # flux.py:164-174
    return _lin(sd, k + ".lin", F.silu(vec))[:, None, :].chunk(n, dim=-1)


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def double_block(sd, b, heads, img, txt, vec, pe):  # flux.py:206-264
    is1, isc1, ig1, is2, isc2, ig2 = _modulation(sd, b + ".img_mod", vec, 6)
    ts1, tsc1, tg1, ts2, tsc2, tg2 = _modulation(sd, b + ".txt_mod", vec, 6)
    iq, ik, iv = _split_qkv(_lin(sd, b + ".img_attn.qkv", (1 + isc1) * _ln(img) + is1), heads)
    iq, ik = _rms(iq, sd[b + ".img_attn.norm.query_norm.scale"]), _rms(ik, sd[b + ".img_attn.norm.key_norm.scale"])
    tq, tk, tv = _split_qkv(_lin(sd, b + ".txt_attn.qkv", (1 + tsc1) * _ln(txt) + ts1), heads)
    tq, tk = _rms(tq, sd[b + ".txt_attn.norm.query_norm.scale"]), _rms(tk, sd[b + ".txt_attn.norm.key_norm.scale"])
    attn = _attention(torch.cat((tq, iq), 2), torch.cat((tk, ik), 2), torch.cat((tv, iv), 2), pe)
    ta, ia = attn[:, :txt.shape[1]], attn[:, txt.shape[1]:]
    img = img + ig1 * _lin(sd, b + ".img_attn.proj", ia)
    img = img + ig2 * _lin(sd, b + ".img_mlp.2", F.gelu(_lin(sd, b + ".img_mlp.0", (1 + isc2) * _ln(img) + is2), approximate="tanh"))
    txt = txt + tg1 * _lin(sd, b + ".txt_attn.proj", ta)
    txt = txt + tg2 * _lin(sd, b + ".txt_mlp.2", F.gelu(_lin(sd, b + ".txt_mlp.0", (1 + tsc2) * _ln(txt) + ts2), approximate="tanh"))
    return img, txt


def single_block(sd, b, heads, hidden, x, vec, pe):  # flux.py:283-307
    shift, scale, gate = _modulation(sd, b + ".modulation", vec, 3)
    h = _lin(sd, b + ".linear1", (1 + scale) * _ln(x) + shift)
    qkv, mlp = h[..., :3 * hidden], h[..., 3 * hidden:]
    q, k, v = _split_qkv(qkv, heads)
    q, k = _rms(q, sd[b + ".norm.query_norm.scale"]), _rms(k, sd[b + ".norm.key_norm.scale"])
    attn = _attention(q, k, v, pe)
    return x + gate * _lin(sd, b + ".linear2", torch.cat((attn, F.gelu(mlp, approximate="tanh")), 2))


def patchify(x):  # flux.py:400-406 (h, w even; circular pad otherwise)
    bs, c, h, w = x.shape
    pad_h, pad_w = (2 - h % 2) % 2, (2 - w % 2) % 2
    x = F.pad(x, (0, pad_w, 0, pad_h), mode="circular")
    hh, ww = x.shape[-2] // 2, x.shape[-1] // 2
    img = x.view(bs, c, hh, 2, ww, 2).permute(0, 2, 4, 1, 3, 5).reshape(bs, hh * ww, c * 4)
    return img, hh, ww


def image_ids(bs, h_len, w_len):  # flux.py:407-413
    ids = torch.zeros(h_len, w_len, 3)
    ids[..., 1] += torch.linspace(0, h_len - 1, steps=h_len)[:, None]
    ids[..., 2] += torch.linspace(0, w_len - 1, steps=w_len)[None, :]
    return ids.reshape(1, h_len * w_len, 3).repeat(bs, 1, 1)


def flux_forward(sd, cfg, x, timestep, context, y, guidance=None):
    """sd: fp32 state dict; x [B,16,h,w]; timestep [B] (= sigma in (0,1]); context [B,Lt,ctx]; y [B,vec]; guidance [B]."""
    sd = {k: v.float() for k, v in sd.items()}
    heads, hidden = cfg["num_heads"], cfg["hidden_size"]
    bs, c, h, w = x.shape
    img, h_len, w_len = patchify(x.float())
    img_ids = image_ids(bs, h_len, w_len)
    txt_ids = torch.zeros(bs, context.shape[1], 3)
    # inner_forward (flux.py:372-398)
    img = _lin(sd, "img_in", img)
    vec = _mlp_embedder(sd, "time_in", timestep_embedding(timestep, 256))
    if cfg["guidance_embed"]:
        vec = vec + _mlp_embedder(sd, "guidance_in", timestep_embedding(guidance, 256))
    vec = vec + _mlp_embedder(sd, "vector_in", y.float())
    txt = _lin(sd, "txt_in", context.float())
    pe = embed_nd(torch.cat((txt_ids, img_ids), 1), cfg["axes_dim"], cfg["theta"])
    for i in range(cfg["depth"]):
        img, txt = double_block(sd, f"double_blocks.{i}", heads, img, txt, vec, pe)
    xj = torch.cat((txt, img), 1)
    for i in range(cfg["depth_single_blocks"]):
        xj = single_block(sd, f"single_blocks.{i}", heads, hidden, xj, vec, pe)
    img = xj[:, txt.shape[1]:]
    shift, scale = _lin(sd, "final_layer.adaLN_modulation.1", F.silu(vec)).chunk(2, dim=1)  # flux.py:317-328
    out = _lin(sd, "final_layer.linear", (1 + scale[:, None, :]) * _ln(img) + shift[:, None, :])
    out = out.view(bs, h_len, w_len, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(bs, c, h_len * 2, w_len * 2)
    return out[:, :, :h, :w]


def flux_sigmas_simple(n, sigma_table):
    """modules/sd_schedulers.py simple_scheduler: sigmas[-(1 + int(x * len/n))] for x in range(n), then 0."""
    ss = len(sigma_table) / n
    return torch.tensor([float(sigma_table[-(1 + int(x * ss))]) for x in range(n)] + [0.0])


def flux_sigma_table(seq_len=4096, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15, pseudo_timestep_range=10000):
    """PredictionFlux.apply_mu_transform (k_prediction.py:291-301)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    mu = seq_len * m + (base_shift - m * base_seq_len)
    t = torch.arange(1, pseudo_timestep_range + 1, 1) / pseudo_timestep_range
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)


def flux_sample_euler(sd, cfg, x, sigmas, context, y, guidance):
    """k_diffusion sample_euler (sampling.py:120-137, s_churn = 0) over KModel.apply_model with prediction_type 'const'
    (k_prediction.py:74-92: input = x, denoised = x - out * sigma)."""
    for i in range(len(sigmas) - 1):
        s = sigmas[i].expand(x.shape[0])
        out = flux_forward(sd, cfg, x, s, context, y, guidance)
        denoised = x - out * sigmas[i]
        d = (x - denoised) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
    return x
