"""TEST INFRASTRUCTURE ONLY (CPU oracle): the Flux transformer with rounding at the native executor's storage sites, in the executor's element type.

The companion of oracle/unet_fp16sites.py / oracle/vae_fp16sites.py for SURVEY row a17 (reference backend/nn/flux.py:206-307 DoubleStreamBlock /
SingleStreamBlock, :372-398 inner_forward, :317-328 LastLayer).  The arithmetic is oracle/flux.py's (pinned to the reference's fixtures,
tests/test_oracle_golden.py); `dtype=None` reproduces it to fp32 summation order.  With `dtype` = torch.bfloat16 (what the reference runs Flux in) or
torch.float16, every tensor the native executor (stable-diffusion-webui-forge_amd/backend/nn/flux.py) STORES is rounded to that type where it stores it:

  * every parameter; the patchified latent, the text context, y; sinusoidal embeddings; each Linear / SiLU of the three MLP embedders, the running sum `vec`
    (added as the residual of the embedder's second GEMM: one rounding), SiLU(vec), the ONE modulation GEMM's output (all adaLN chunks are slices of it);
  * img_in / txt_in outputs; per block: (1 + scale) LayerNorm(x) + shift (one rounding, fp32 statistics of the stored x), the qkv GEMM's output, q and k
    after RMS-norm x scale and RoPE (one rounding, fp32 in between), V as copied; the attention output -- scores in fp32 from the UNSCALED q (the 128-wide
    kernel applies scale x log2 e to the fp32 scores), P rounded for P V, the row sum over the unrounded exponentials; x + gate (W a + b) with one rounding
    (gate and residual live in the GEMM epilogue); GELU-tanh(W x + b); the single-stream block's linear2 over [attention | mlp] (two K sources, one GEMM);
  * the final adaLN + Linear.

`teacher` / `layer_out`: as in the UNet / VAE oracles -- every stage evaluated on the NATIVE output of the stage in front of it.  Keys: "vec",
"img_in", "txt_in"; per double block b: "b.q", "b.k", "b.v" (joint text|image sequence, [B, L, hidden]), "b.attn", "b.<st>.a" (stream after the attention
residual), "b.<st>.h" (GELU output), "b.<st>" (stream after the MLP); per single block: "b.q/.k/.v", "b.mlp", "b.attn", "b"; "out" ([B, 16, h, w]).
`plant`: {"ln_eps": (block key, eps)} evaluates ONE block's first adaLN LayerNorm with another epsilon; {"gelu_erf": block key} uses the exact GELU in ONE
block's MLP (planted bugs).  parity: pinned through oracle/flux.py and the reference's own bf16 / fp16 floors (tests/test_oracle_flux_sites.py)."""
import torch
import torch.nn.functional as F

from . import flux as of

LOG2E = 1.4426950408889634


class _State:
    def __init__(self, dtype, teacher, layer_out, plant):
        self.dtype, self.teacher, self.layer_out, self.plant = dtype, teacher, layer_out, plant or {}
        self.R = (lambda t: t.to(dtype).float()) if dtype is not None else (lambda t: t)

    def teach(self, key, computed):
        if self.layer_out is not None:
            self.layer_out[key] = computed
        if self.teacher is not None and key in self.teacher:
            return self.teacher[key].float().reshape(computed.shape)
        return computed


def _lin(st, sd, k, x):
    b = sd.get(k + ".bias")
    return F.linear(x, st.R(sd[k + ".weight"]), None if b is None else st.R(b))


def _embed(st, sd, k, x, residual=None):
    """MLPEmbedder (flux.py:94-104) as the executor runs it: Linear, SiLU in place, Linear (+ the running vec as the GEMM's residual)"""
    R = st.R
    h = R(F.silu(R(_lin(st, sd, k + ".in_layer", x))))
    o = _lin(st, sd, k + ".out_layer", h)
    return R(o if residual is None else o + residual)


def _ln_mod(st, x, scale, shift, eps=1e-6):
    """(1 + scale) LayerNorm(x) + shift, one rounding"""
    return st.R((1 + scale) * F.layer_norm(x, (x.shape[-1],), eps=eps) + shift)


def _rms_rope(st, t, scale, pe):
    """q or k [B, H, L, D]: RMS-norm x scale (flux.py:107-126), RoPE (:43-49), one rounding"""
    t = t * torch.rsqrt(torch.mean(t * t, dim=-1, keepdim=True) + 1e-6) * st.R(scale)
    tp = t.reshape(*t.shape[:-1], -1, 1, 2)
    return st.R((pe[..., 0] * tp[..., 0] + pe[..., 1] * tp[..., 1]).reshape(t.shape))


def _attention(st, q, k, v):
    """q, k, v [B, H, L, D] as stored -> [B, L, H D]"""
    d = q.shape[-1]
    b, h, l, _ = q.shape
    if st.dtype is None:
        s = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
        o = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)
    else:
        c2 = torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)
        o = torch.empty_like(q)
        step = max(1, (1 << 25) // max(1, l * h))
        for i in range(0, l, step):
            s = torch.matmul(q[:, :, i:i + step], k.transpose(-1, -2)) * c2
            p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
            o[:, :, i:i + step] = torch.matmul(st.R(p), v) / p.sum(dim=-1, keepdim=True)
        o = st.R(o)
    return o.permute(0, 2, 1, 3).reshape(b, l, h * d)


def _heads(t, heads):
    b, l, c = t.shape
    return t.view(b, l, heads, c // heads).permute(0, 2, 1, 3)


def _unheads(t):
    b, h, l, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, l, h * d)


def _gelu(st, x, key):
    return F.gelu(x, approximate="none" if st.plant.get("gelu_erf") == key else "tanh")


def _mod_chunks(st, sd, k, svec, n):
    """the adaLN chunks of one Modulation (flux.py:164-174): slices of the executor's one modulation GEMM = this Linear on SiLU(vec), rounded"""
    return st.R(_lin(st, sd, k, svec))[:, None, :].chunk(n, dim=-1)


def double_block(st, sd, b, heads, img, txt, svec, pe, lt):
    R = st.R
    hs = img.shape[-1]
    mods = {"img": _mod_chunks(st, sd, b + ".img_mod.lin", svec, 6), "txt": _mod_chunks(st, sd, b + ".txt_mod.lin", svec, 6)}
    cur = {"img": img, "txt": txt}
    q, k, v = {}, {}, {}
    p = st.plant.get("ln_eps")
    for s in ("txt", "img"):
        sh, sc = mods[s][0], mods[s][1]
        xm = _ln_mod(st, cur[s], sc, sh, p[1] if p and p[0] == b and s == "img" else 1e-6)
        qkv = R(_lin(st, sd, f"{b}.{s}_attn.qkv", xm))
        qq, kk, vv = of._split_qkv(qkv, heads)
        off = slice(0, lt) if s == "txt" else slice(lt, None)
        q[s] = _rms_rope(st, qq, sd[f"{b}.{s}_attn.norm.query_norm.scale"], pe[:, :, off])
        k[s] = _rms_rope(st, kk, sd[f"{b}.{s}_attn.norm.key_norm.scale"], pe[:, :, off])
        v[s] = vv
    qj = _heads(st.teach(b + ".q", _unheads(torch.cat((q["txt"], q["img"]), 2))), heads)
    kj = _heads(st.teach(b + ".k", _unheads(torch.cat((k["txt"], k["img"]), 2))), heads)
    vj = _heads(st.teach(b + ".v", _unheads(torch.cat((v["txt"], v["img"]), 2))), heads)
    attn = st.teach(b + ".attn", _attention(st, qj, kj, vj))
    out = {}
    for s in ("img", "txt"):
        a = attn[:, :lt] if s == "txt" else attn[:, lt:]
        g1, sh2, sc2, g2 = mods[s][2], mods[s][3], mods[s][4], mods[s][5]
        x = st.teach(f"{b}.{s}.a", R(cur[s] + g1 * _lin(st, sd, f"{b}.{s}_attn.proj", a)))
        hdn = st.teach(f"{b}.{s}.h", R(_gelu(st, _lin(st, sd, f"{b}.{s}_mlp.0", _ln_mod(st, x, sc2, sh2)), b)))
        out[s] = st.teach(f"{b}.{s}", R(x + g2 * _lin(st, sd, f"{b}.{s}_mlp.2", hdn)))
    return out["img"], out["txt"]


def single_block(st, sd, b, heads, hidden, x, svec, pe):
    R = st.R
    shift, scale, gate = _mod_chunks(st, sd, b + ".modulation.lin", svec, 3)
    p = st.plant.get("ln_eps")
    xm = _ln_mod(st, x, scale, shift, p[1] if p and p[0] == b else 1e-6)
    w1, b1 = st.R(sd[b + ".linear1.weight"]), st.R(sd[b + ".linear1.bias"])
    qkv = R(F.linear(xm, w1[:3 * hidden], b1[:3 * hidden]))
    mlp = st.teach(b + ".mlp", R(_gelu(st, F.linear(xm, w1[3 * hidden:], b1[3 * hidden:]), b)))
    qq, kk, vv = of._split_qkv(qkv, heads)
    qj = _heads(st.teach(b + ".q", _unheads(_rms_rope(st, qq, sd[b + ".norm.query_norm.scale"], pe))), heads)
    kj = _heads(st.teach(b + ".k", _unheads(_rms_rope(st, kk, sd[b + ".norm.key_norm.scale"], pe))), heads)
    vj = _heads(st.teach(b + ".v", _unheads(vv)), heads)
    attn = st.teach(b + ".attn", _attention(st, qj, kj, vj))
    return st.teach(b, R(x + gate * _lin(st, sd, b + ".linear2", torch.cat((attn, mlp), 2))))


@torch.no_grad()
def flux_forward(sd, cfg, x, timestep, context, y, guidance=None, dtype=torch.bfloat16, teacher=None, layer_out=None, plant=None):
    """sd: fp32 state dict; x [B,16,h,w]; timestep [B]; context [B,Lt,ctx]; y [B,vec]; guidance [B] -> [B,16,h,w] (values of `dtype`, as fp32)."""
    st = _State(dtype, teacher, layer_out, plant)
    R = st.R
    sd = {k: v.float() for k, v in sd.items()}
    heads, hidden = cfg["num_heads"], cfg["hidden_size"]
    bs, c, h, w = x.shape
    img, h_len, w_len = of.patchify(R(x.float()))
    lt = context.shape[1]
    ids = torch.cat((torch.zeros(bs, lt, 3), of.image_ids(bs, h_len, w_len)), 1)
    pe = of.embed_nd(ids, cfg["axes_dim"], cfg["theta"])                    # [B, 1, L, D/2, 2, 2] fp32 (the executor's table is fp32 too)
    vec = _embed(st, sd, "time_in", R(of.timestep_embedding(timestep, 256)))
    if cfg["guidance_embed"]:
        vec = _embed(st, sd, "guidance_in", R(of.timestep_embedding(guidance, 256)), residual=vec)
    vec = st.teach("vec", _embed(st, sd, "vector_in", R(y.float()), residual=vec))
    svec = R(F.silu(vec))
    img = st.teach("img_in", R(_lin(st, sd, "img_in", img)))
    txt = st.teach("txt_in", R(_lin(st, sd, "txt_in", R(context.float()))))
    for i in range(cfg["depth"]):
        img, txt = double_block(st, sd, f"double_blocks.{i}", heads, img, txt, svec, pe, lt)
    xj = torch.cat((txt, img), 1)
    for i in range(cfg["depth_single_blocks"]):
        xj = single_block(st, sd, f"single_blocks.{i}", heads, hidden, xj, svec, pe)
    img = xj[:, lt:]
    shift, scale = R(_lin(st, sd, "final_layer.adaLN_modulation.1", svec)).chunk(2, dim=1)
    out = R(_lin(st, sd, "final_layer.linear", _ln_mod(st, img, scale[:, None, :], shift[:, None, :])))
    out = out.view(bs, h_len, w_len, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(bs, c, h_len * 2, w_len * 2)
    return st.teach("out", out[:, :, :h, :w])


def kind_of(key):
    if key in ("vec", "img_in", "txt_in", "out"):
        return key
    tail = key.rsplit(".", 1)[-1]
    if tail in ("q", "k", "v"):
        return "q / k (norm + RoPE) / v"
    if tail == "attn":
        return "attention output"
    if tail in ("h", "mlp"):
        return "GELU output"
    if tail == "a":
        return "stream after the attention residual"
    return "block output"
