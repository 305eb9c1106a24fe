"""Attention-level drop-in: `attention_function` and `attention_function_single_head_spatial` with the reference's signatures
(backend/attention.py:324-339 `attention_pytorch`, :37-93 `attention_basic`, :412-422 `pytorch_attention_single_head_spatial`; selection
:430-451), executed by the fused MI355X kernel behind `fmx_attention_f16`.

A Forge install binds here with one assignment each:

    backend.attention.attention_function = forge_amd.backend.attention.attention_function
    backend.attention.attention_function_single_head_spatial = forge_amd.backend.attention.attention_function_single_head_spatial

The reference hands over `[B, N, heads*dim_head]` (or `[B, heads, N, dim_head]` with skip_reshape) tensors of any float type and gets
`[B, Nq, heads*dim_head]` back in q's dtype.  The kernel wants fp16 heads zero-padded to an MFMA-friendly width and V transposed; the
adapters are ONE strided-copy kernel (`fmx_strided_copy4`: conversion, head padding, V -> V^T, bool mask -> additive mask) -- nothing
here falls back to torch arithmetic, and without libfmx_gfx950.so every call raises.  The native UNet / VAE executors do not pay for
these adapters: their projection GEMMs write the kernel's layouts directly (backend/nn/unet.py).

`mask` (both forms the reference accepts): a bool mask whose True entries attend, `[B, Nk]` (attention_basic, :74-78) or anything
broadcastable to `[B, heads, Nq, Nk]` (SDPA); an additive float mask `[Nq, Nk]`, `[bs, Nq, Nk]` (attention_basic :79-85) or broadcastable to
`[B, heads, Nq, Nk]`.  A 2-D bool mask is the `[B, Nk]` key mask when its shape says so; when B == Nq makes that ambiguous the call raises.
`attn_precision` is accepted and ignored: scores, softmax statistics and the output accumulate in fp32 always -- but q, k, v (and the
probabilities fed to P.V) are fp16 OPERANDS here, where the reference's fp32-upcast path (`attn_precision = torch.float32`, :45-47, :64-67) would
carry them in fp32; bf16 / fp32 inputs are converted to fp16 by the adapter (values beyond 65504 overflow: the UNet / VAE activations the
reference feeds this function are fp16 already).
"""
import torch

from .. import hipops as ops

SUPPORTED_DPAD = (48, 64, 80, 128, 160)


def _dpad(d):
    for s in SUPPORTED_DPAD:
        if d <= s:
            return s
    return None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise TypeError("forge_amd.backend.attention runs on the MI355X only: got a tensor on %s (there is no CPU path)" % t.device)


def _mask_view(mask, b, heads, nq, nk):
    """The mask as a 4-D [B | 1, heads | 1, Nq | 1, Nk] view, by the reference's reading of each rank (host logic only: no kernel)."""
    m = mask
    if m.dtype == torch.bool and m.dim() == 2 and m.shape == (b, nk) and nq == b and b > 1:
        # [B, Nk] key mask (attention_basic :74-78) and [Nq, Nk] SDPA mask have the same shape here and mean different things.  The reference's
        # explicit form ALWAYS reads a 2-D bool mask as the per-batch key mask ('b ... -> b (...)'), so that is what it means here too; a caller
        # who means the per-query reading passes [1, 1, Nq, Nk] (round 3 raised here, which broke callers following the reference contract)
        import warnings
        warnings.warn(f"2-D bool attention mask of shape {tuple(m.shape)} with B == Nq == {b}: read as the reference's [B, Nk] key mask "
                      f"(attention_basic); pass [1, 1, Nq, Nk] for a per-query mask", stacklevel=3)
    if m.dtype == torch.bool and m.dim() >= 2 and m.shape[0] == b and m.dim() != 4 and m[0].numel() == nk and not (m.dim() == 2 and nq == b == 1):
        m = m.reshape(b, 1, 1, nk)                                    # attention_basic's 'b ... -> b (...)' key mask
    elif m.dim() == 2:
        m = m.reshape(1, 1, m.shape[0], m.shape[1])
    elif m.dim() == 3:
        m = m.reshape(m.shape[0], 1, m.shape[1], m.shape[2])
    elif m.dim() != 4:
        raise ValueError(f"attention mask of shape {tuple(mask.shape)} is not broadcastable to [B, heads, Nq, Nk]")
    mb, mh, mq, mk = m.shape
    if mk != nk or mb not in (1, b) or mh not in (1, heads) or mq not in (1, nq):
        raise ValueError(f"attention mask {tuple(mask.shape)} does not broadcast to [{b}, {heads}, {nq}, {nk}]")
    if m.dtype not in (torch.bool, torch.float16, torch.float32, torch.bfloat16):
        raise TypeError(f"attention mask dtype {m.dtype}")
    return m


def _additive_mask(mask, b, heads, nq, nk, nkp, device):
    """-> (fp16 tensor with rows of nkp keys, (batch, head, query) element strides) for the kernel."""
    m = _mask_view(mask, b, heads, nq, nk)
    mb, mh, mq, mk = m.shape
    out = torch.zeros(mb, mh, mq, nkp, dtype=torch.float16, device=device)
    ops.strided_copy4(m, out, (mb, mh, mq, nk), m.stride(), out.stride())
    strides = (out.stride(0) if mb > 1 else 0, out.stride(1) if mh > 1 else 0, out.stride(2) if mq > 1 else 0)
    return out, strides


def attention_function(q, k, v, heads, mask=None, attn_precision=None, skip_reshape=False):
    """Reference: backend/attention.py:324-339 (and the explicit form :37-93).  q [B, Nq, heads*d] / k, v [B, Nk, heads*d]
    (skip_reshape: [B, heads, N, d]) -> [B, Nq, heads*d] in q's dtype."""
    _need_cuda(q, k, v, mask)
    if skip_reshape:
        b, h_, nq, d = q.shape
        assert h_ == heads
        nk = k.shape[2]
        st = lambda t: (t.stride(0), t.stride(2), t.stride(1), t.stride(3))      # -> (b, n, h, d) strides
    else:
        b, nq, hd = q.shape
        d = hd // heads
        nk = k.shape[1]
        st = lambda t: (t.stride(0), t.stride(1), d * t.stride(2), t.stride(2))
    dp = _dpad(d)
    if dp is None:
        if heads != 1:
            raise NotImplementedError(f"head dim {d} > 160 with {heads} heads: the fused kernel covers d_head <= 160, the single-head form any width")
        if mask is not None:
            raise NotImplementedError(f"attention mask with a {d}-wide single head: the wide single-head kernels take no mask (the VAE passes none)")
        o = _single_head_tokens(q.reshape(b, nq, d), k.reshape(b, nk, d), v.reshape(b, nk, d))
        return o if not skip_reshape else o
    dev = q.device
    nkp = -(-nk // 64) * 64
    qh = torch.zeros(b, nq, heads, dp, dtype=torch.float16, device=dev) if dp != d else torch.empty(b, nq, heads, dp, dtype=torch.float16, device=dev)
    kh = torch.zeros(b, nkp, heads, dp, dtype=torch.float16, device=dev)
    vt = torch.zeros(heads, dp, b, nkp, dtype=torch.float16, device=dev)         # V^T[(h, d)][b * nkp + j]
    ops.strided_copy4(q, qh, (b, nq, heads, d), st(q), qh.stride())
    ops.strided_copy4(k, kh, (b, nk, heads, d), st(k), kh.stride())
    ops.strided_copy4(v, vt, (b, nk, heads, d), st(v), (vt.stride(2), vt.stride(3), vt.stride(0), vt.stride(1)))
    m, ms = (None, (0, 0, 0)) if mask is None else _additive_mask(mask, b, heads, nq, nk, nkp, dev)
    o = ops.attention(qh, kh, vt, batch=b, heads=heads, nq=nq, nk=nk, nk_pad=nkp, dpad=dp, scale=d ** -0.5, q_bs=nq * heads * dp, q_rs=heads * dp,
                      k_bs=nkp * heads * dp, k_rs=heads * dp, vt_bs=nkp, vt_hs=dp * b * nkp, vt_ds=b * nkp, mask=m, mask_strides=ms)
    out = torch.empty(b, nq, heads * d, dtype=q.dtype, device=dev)
    o4 = o.view(b, nq, heads, dp)
    ops.strided_copy4(o4, out, (b, nq, heads, d), o4.stride(), (nq * heads * d, heads * d, d, 1))
    return out


def _single_head_tokens(q, k, v):
    """One head of arbitrary width (the VAE mid block: 512 channels, up to 16 384 tokens): [B, N, C] each -> [B, Nq, C] in q's dtype.
    Widths the fused kernel covers go through it; wider heads run S = scale * Q K^T (MFMA GEMM) -> row softmax -> P V (MFMA GEMM) per
    image, as the native VAE executor does (backend/nn/vae.py)."""
    b, nq, c = q.shape
    nk = k.shape[1]
    if _dpad(c) is not None:
        return attention_function(q, k, v, 1)
    if c % 64:
        raise NotImplementedError(f"single-head attention needs a channel count that is a multiple of 64 beyond 160, got {c}")
    dev = q.device
    npad = -(-nk // 64) * 64
    out = torch.empty(b, nq, c, dtype=q.dtype, device=dev)
    if c == 512:   # the VAE's width: fused kernel, whole batch in one launch (csrc/fmx_attention512.hip)
        q2 = torch.empty(b * nq, c, dtype=torch.float16, device=dev)
        k2 = torch.empty(b * nk, c, dtype=torch.float16, device=dev)
        vt = torch.zeros(c, b * npad, dtype=torch.float16, device=dev)
        ops.strided_copy4(q, q2, (1, b, nq, c), (0, q.stride(0), q.stride(1), q.stride(2)), (0, nq * c, c, 1))
        ops.strided_copy4(k, k2, (1, b, nk, c), (0, k.stride(0), k.stride(1), k.stride(2)), (0, nk * c, c, 1))
        ops.strided_copy4(v, vt, (1, b, nk, c), (0, v.stride(0), v.stride(1), v.stride(2)), (0, npad, 1, b * npad))
        o2 = torch.empty(b * nq, c, dtype=torch.float16, device=dev)
        ops.attention_single_head512(q2, k2, vt, o2, batch=b, nq=nq, nk=nk, nk_pad=npad, q_bs=nq * c, q_rs=c, k_bs=nk * c, k_rs=c, vt_bs=npad,
                                     vt_ds=b * npad, scale=c ** -0.5)
        ops.strided_copy4(o2, out, (1, b, nq, c), (0, nq * c, c, 1), (0, nq * c, c, 1))
        return out
    qi = torch.empty(nq, c, dtype=torch.float16, device=dev)
    ki = torch.empty(nk, c, dtype=torch.float16, device=dev)
    vt = torch.zeros(c, npad, dtype=torch.float16, device=dev)
    s = torch.zeros(nq, npad, dtype=torch.float16, device=dev)
    o = torch.empty(nq, c, dtype=torch.float16, device=dev)
    for bi in range(b):
        ops.strided_copy4(q[bi], qi, (1, 1, nq, c), (0, 0, q.stride(1), q.stride(2)), (0, 0, c, 1))
        ops.strided_copy4(k[bi], ki, (1, 1, nk, c), (0, 0, k.stride(1), k.stride(2)), (0, 0, c, 1))
        ops.strided_copy4(v[bi], vt, (1, 1, nk, c), (0, 0, v.stride(1), v.stride(2)), (0, 0, 1, npad))
        ops.conv_gemm(qi, ki, nk, alpha=c ** -0.5, out=s, ld_out=npad)
        ops.softmax_rows_(s[:, :nk])
        if npad != nk:
            s[:, nk:].zero_()
        ops.conv_gemm(s, vt, c, out=o, ld_out=c)
        ops.strided_copy4(o, out[bi], (1, 1, nq, c), (0, 0, c, 1), (0, 0, c, 1))
    return out


def attention_function_single_head_spatial(q, k, v):
    """Reference: backend/attention.py:412-422 -- q, k, v [B, C, H, W], one head of width C -> [B, C, H, W] (the VAE mid-block attention)."""
    _need_cuda(q, k, v)
    b, c, hh, ww = q.shape
    n = hh * ww
    tok = lambda t: t.reshape(b, c, n).transpose(1, 2)                             # [B, N, C] view of NCHW: no copy
    o = _single_head_tokens(tok(q), tok(k), tok(v))                               # [B, N, C]
    out = torch.empty(b, c, hh, ww, dtype=q.dtype, device=q.device)
    ops.strided_copy4(o, out, (1, b, n, c), (0, n * c, c, 1), (0, c * n, 1, n))
    return out
