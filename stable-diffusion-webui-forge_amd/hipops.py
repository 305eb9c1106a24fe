"""Functional wrappers: torch tensors (device memory + streams only) -> C-ABI calls of libfmx_gfx950.so.

Activations are fp16 NHWC: a feature map is a contiguous [N, H, W, C] tensor and -- for free -- the token
matrix [N*H*W, C] the transformer blocks want (the reference pays two transposes per SpatialTransformer,
backend/nn/unet.py:315,324).  Every function launches on torch's CURRENT stream and returns immediately.
Temporary / output buffers come from the active `Arena` (runtime.py) when one is installed, so a whole UNet
forward allocates nothing from HIP and can be captured into a HIP graph.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import AttnArgs, ConvGnArgs, GemmArgs

ACT_NONE, ACT_GEGLU, ACT_GELU_TANH = 0, 1, 2

_zero_pages = {}
_alloc = None  # callable(shape, dtype) -> tensor ; set by runtime.Arena
_profiler = None  # KernelProfiler while bench.py measures per-launch kernel time


class KernelProfiler:
    """Brackets every launch of the two MFMA kernels with HIP events recorded on the launch stream (fmx_event_*), so
    bench.py can report algorithmic FLOP / measured kernel time per kernel family (roofline.achieved)."""

    def __init__(self):
        self.records = []  # (kind, flops, ev_start, ev_stop)
        self.tags = []     # problem shape per record (bench.py --breakdown)
        self.by_tag = {}

    def __enter__(self):
        global _profiler
        self._prev, _profiler = _profiler, self
        return self

    def __exit__(self, *exc):
        global _profiler
        _profiler = self._prev
        return False

    def _event(self):
        ev = C.c_void_p()
        _lib.check(_lib.lib().fmx_event_create(C.byref(ev)), "fmx_event_create")
        return ev

    def launch(self, kind, flops, fn, tag=None, nbytes=0.0):
        """nbytes: ALGORITHMIC HBM bytes of the bracketed launches (HBM-bound kernels: GroupNorm = 1 read + 1 write of the tensor)."""
        L = _lib.lib()
        a, b = self._event(), self._event()
        sp = stream_ptr()
        _lib.check(L.fmx_event_record(a, sp), "fmx_event_record")
        fn()
        _lib.check(L.fmx_event_record(b, sp), "fmx_event_record")
        self.records.append((kind, flops, a, b, nbytes))
        self.tags.append(tag)

    def summary(self):
        """-> {kind: {"launches", "flops", "seconds"}} ; synchronises."""
        L = _lib.lib()
        out = {}
        for (kind, flops, a, b, nbytes), tag in zip(self.records, self.tags):
            ms = C.c_float()
            _lib.check(L.fmx_event_elapsed_ms(a, b, C.byref(ms)), "fmx_event_elapsed_ms")
            for d in (out.setdefault(kind, {"launches": 0, "flops": 0.0, "seconds": 0.0, "bytes": 0.0}),
                      self.by_tag.setdefault((kind, tag), {"launches": 0, "flops": 0.0, "seconds": 0.0, "bytes": 0.0})):
                d["launches"] += 1
                d["flops"] += flops
                d["bytes"] += nbytes
                d["seconds"] += ms.value * 1e-3
            L.fmx_event_destroy(a)
            L.fmx_event_destroy(b)
        self.records, self.tags = [], []
        return out


# developer hook: FMX_DEBUG_NAN=1 synchronises after every GEMM / norm / attention launch (outside graph capture) and raises at the first
# non-finite output -- pinpoints the producing kernel; never set in production (it serialises the stream)
_DEBUG_NAN = _lib.knob("FMX_DEBUG_NAN") == "1"


def _dbg(what, **tensors):
    if not _DEBUG_NAN or torch.cuda.is_current_stream_capturing():
        return
    torch.cuda.synchronize()
    for k, t in tensors.items():
        if t is not None and not bool(torch.isfinite(t.float()).all()):
            bad = (~torch.isfinite(t.float())).nonzero()
            raise FloatingPointError(f"{what}: {k} {tuple(t.shape)} has {bad.shape[0]} non-finite values, first at {bad[0].tolist()}")


def set_allocator(fn):
    global _alloc
    prev = _alloc
    _alloc = fn
    return prev


def empty(shape, dtype=torch.float16, device=None):
    if _alloc is not None:
        return _alloc(tuple(int(s) for s in shape), dtype)
    return torch.empty(shape, dtype=dtype, device=device or torch.device("cuda", torch.cuda.current_device()))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def zero_page(device):
    key = (device.type, device.index)
    zp = _zero_pages.get(key)
    if zp is None:
        zp = torch.zeros(4096, dtype=torch.uint8, device=device)
        _zero_pages[key] = zp
    return zp


SPLITK_WORKSPACE_BYTES = 64 << 20
_splitk_ws = {}


def splitk_workspace(device):
    """The split-K workspace of fmx_gemm_args (include/fmx.h): one per device, arrival counters zeroed once, alive for the process -- a
    captured graph keeps its address.  All launches go to ONE stream at a time (the executors' discipline), as the ABI asks."""
    key = (device.type, device.index)
    ws = _splitk_ws.get(key)
    if ws is None:
        ws = torch.zeros(SPLITK_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _splitk_ws[key] = ws
    return ws


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _check_f16(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.float16 or not t.is_cuda):
            raise TypeError("expected fp16 device tensors, got %s on %s" % (t.dtype, t.device))


def _elem(*ts):
    """-> (symbol suffix, torch dtype) of a 16-bit kernel call: all tensors fp16 (the SD / SDXL path and the default everywhere) or all
    bfloat16 (the Flux executor's optional compute type; only the entry points with a _bf16 build accept it)."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype not in (torch.float16, torch.bfloat16):
            raise TypeError("expected fp16 / bf16 device tensors, got %s on %s" % (t.dtype, t.device))
        if dt is None:
            dt = t.dtype
        elif t.dtype != dt:
            raise TypeError("mixed fp16 / bf16 operands in one kernel call")
    return ("_bf16", torch.bfloat16) if dt == torch.bfloat16 else ("_f16", torch.float16)


class GnStats:
    """GroupNorm statistics of one NHWC fp16 tensor: partial[n][nchunks][c][2] fp32 = {sum, sum of squares} per chunk of pixels."""
    __slots__ = ("partial", "nchunks")

    def __init__(self, partial, nchunks):
        self.partial, self.nchunks = partial, nchunks


# A/B knob (tools/bench_kernels.py, bench.py --breakdown): FMX_GN_FUSED_STATS=0 makes conv_gemm(stats=True) return no statistics, so that
# every GroupNorm runs its own statistics pass as in round 1
_FUSED_STATS = _lib.knob("FMX_GN_FUSED_STATS", "1") != "0"


def conv_gemm(x, wgt, nout, *, x1=None, n=None, h=None, w=None, kh=1, stride=1, pad=0, up=None, bias=None,
              rowvec=None, residual=None, act=ACT_NONE, alpha=1.0, out=None, ld_out=None, out_dtype=None,
              ldw=0, force_tile=0, gate=None, out_hw=None, stats=False, stats_partial=None, row_stats=None, ln=None, ln_swapped=None, ln_ab_out=None,
              xattn=None):
    """OUT[M, ncols] = epilogue(A (*) W^T).  x: [N,H,W,C0] (or [M,C0] with kh == 1); x1: optional second source
    concatenated along channels; wgt: [nout, kh*kh*(C0+C1)]; up=(UH, UW): nearest-resize before the conv.
    stats=True: returns (out, GnStats of out) -- the GroupNorm statistics of the output come out of the GEMM's epilogue (256-row tiles)
    or of a pass behind it, fmx_gemm_conv_stats_f16; (out, None) when the knob above disables it.  stats_partial: a buffer from
    `stats_buffer` when the statistics have to outlive the current arena scope (they live exactly as long as `out` must).
    LayerNorm folding (fmx.h, fmx_gemm_linear_rowstats_f16 / ln_* fields): row_stats = a RowStats whose buffer takes the per-row sums of the
    output (its `.parts` is 0 afterwards if the dispatcher did not use the 256x320 tile); ln = (RowStats of the INPUT, colsum fp32 [nout],
    eps) runs the GEMM as `LN(x) W^T + b` on the un-normalised x (wgt / bias pre-folded by the caller);  ln_swapped = (col_ab fp32 [nout, 2] from
    `ln_rowstats_finalize`, row_cb fp32 [M, 2]) is the same fold for the operand-swapped form `W' x^T` (x = the gamma-scaled weight, wgt = the
    un-normalised activations: V^T of self-attention), fmx.h ln_col_ab / ln_row_cb.
    xattn = (k [images * keys_pad, heads * 64], vt [heads * 64, images * keys_pad], nk, keys_pad, queries_per_image, scale) with `ln`: the query
    projection of a cross-attention whose attention runs in the GEMM's epilogue (fmx.h xa_*): `out` receives the attention output, Q is never stored."""
    sfx, elem = _elem(x, x1, wgt, bias, rowvec, residual, gate)
    fn_name = "fmx_gemm_conv" + sfx
    if out_dtype is None:
        out_dtype = elem
    if x.dim() == 2:
        n_, h_, w_ = 1, 1, x.shape[0]
    else:
        n_, h_, w_ = x.shape[0], x.shape[1], x.shape[2]
    n_, h_, w_ = (n or n_), (h or h_), (w or w_)
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    ih, iw = (up if up is not None else (h_, w_))
    oh = (ih + 2 * pad - kh) // stride + 1
    ow = (iw + 2 * pad - kh) // stride + 1
    if out_hw is not None:  # asymmetric padding (VAE Downsample pads right / bottom only, vae.py:67-70): taps beyond the
        oh, ow = out_hw     # input are bounds-checked zeros, so only the output extent changes
    m = n_ * oh * ow
    ncols = nout // 2 if act == ACT_GEGLU else nout
    if out is None:
        out = empty((m, ncols), out_dtype, x.device)
        ld_out = ncols
    elif ld_out is None:
        ld_out = out.stride(-2) if out.dim() >= 2 else ncols
    a = GemmArgs()
    a.a0, a.a1, a.c0, a.c1 = _p(x), _p(x1), c0, c1
    a.a0_stride = x.stride(-2)
    a.a1_stride = x1.stride(-2) if x1 is not None else 0
    a.n, a.h, a.w, a.oh, a.ow = n_, h_, w_, oh, ow
    a.kh, a.stride, a.pad = kh, stride, pad
    a.up_h, a.up_w = (up if up is not None else (0, 0))
    a.wgt, a.ldw, a.nout = _p(wgt), ldw or wgt.stride(0), nout
    a.bias, a.rowvec = _p(bias), _p(rowvec)
    a.ld_rowvec = rowvec.stride(0) if rowvec is not None else 0
    a.residual = _p(residual)
    a.ld_res = residual.stride(-2) if residual is not None else 0
    a.alpha, a.act = float(alpha), act
    a.out, a.ld_out = _p(out), ld_out
    a.out_f32 = -force_tile if force_tile else (1 if out.dtype == torch.float32 else 0)
    a.zero_page = _p(zero_page(x.device))
    a.gate = _p(gate)
    a.ld_gate = gate.stride(0) if gate is not None else 0
    a.workspace, a.workspace_bytes = _p(splitk_workspace(x.device)), SPLITK_WORKSPACE_BYTES
    global LN_FOLDED_LAUNCHES
    if ln is not None:
        LN_FOLDED_LAUNCHES += 1
        rs, colsum, eps = ln
        assert rs.parts >= 2 and colsum.dtype == torch.float32 and colsum.numel() == nout and rs.partial.dtype == torch.float32
        a.ln_partial, a.ln_parts, a.ln_colsum, a.ln_eps = _p(rs.partial), rs.parts, _p(colsum), float(eps)
        if ln_ab_out is not None:   # the {rstd, -mean rstd} pairs of the input rows, for an operand-swapped fold on the same rows (fmx.h ln_ab_out)
            assert ln_ab_out.dtype == torch.float32 and ln_ab_out.numel() == 2 * m
            a.ln_ab_out = _p(ln_ab_out)
    if xattn is not None:
        xk, xvt, xnk, xpad, xrows, xscale = xattn
        assert ln is not None and sfx == "_f16" and xk.dtype == torch.float16 and xvt.dtype == torch.float16 and xk.is_contiguous() and xvt.is_contiguous()
        a.xa_k, a.xa_vt = _p(xk), _p(xvt)
        a.xa_k_rs, a.xa_k_bs, a.xa_vt_ds, a.xa_vt_bs = xk.shape[1], xpad, xvt.shape[1], xpad
        a.xa_nk, a.xa_rows, a.xa_scale = int(xnk), int(xrows), float(xscale)
        a.xa_k_bytes, a.xa_vt_bytes = xk.numel() * 2, xvt.numel() * 2
    if ln_swapped is not None:
        LN_FOLDED_LAUNCHES += 1
        col_ab, row_cb = ln_swapped
        assert col_ab.dtype == torch.float32 and col_ab.numel() == 2 * nout and row_cb.dtype == torch.float32 and row_cb.numel() == 2 * m and ln is None
        a.ln_col_ab, a.ln_row_cb = _p(col_ab), _p(row_cb)
    st = None
    if row_stats is not None:
        assert sfx == "_f16" and not stats and ln is None and ln_swapped is None
        cap = row_stats.partial.numel() // (2 * m)
        got = C.c_int32(0)

        def launch():
            _lib.check(_lib.lib().fmx_gemm_linear_rowstats_f16(C.byref(a), _p(row_stats.partial), cap, C.byref(got), stream_ptr()),
                       "fmx_gemm_linear_rowstats_f16")
            row_stats.parts = got.value
    elif stats and _FUSED_STATS:
        hw = oh * ow
        fb, cap = _stats_geometry(n_, hw)
        partial = stats_partial if stats_partial is not None else empty((n_, cap, ncols, 2), torch.float32, x.device)
        assert partial.dtype == torch.float32 and partial.numel() >= n_ * cap * ncols * 2
        nch = C.c_int32(0)
        st = GnStats(partial, 0)

        def launch():
            _lib.check(getattr(_lib.lib(), "fmx_gemm_conv_stats" + sfx)(C.byref(a), _p(partial), cap, fb, C.byref(nch), stream_ptr()),
                       "fmx_gemm_conv_stats" + sfx)
            st.nchunks = nch.value
    else:
        def launch():
            _lib.check(getattr(_lib.lib(), fn_name)(C.byref(a), stream_ptr()), fn_name)
    if _profiler is not None:
        flops = 2.0 * m * nout * kh * kh * (c0 + c1)
        _profiler.launch("gemm_conv", flops, launch,
                         tag=f"M={m} N={nout} K={kh * kh * (c0 + c1)} kh={kh} s={stride}{' up' if up else ''}{' geglu' if act == ACT_GEGLU else ''}"
                             f"{' +gnstats' if st is not None else ''}{' +xattn' if xattn is not None else ''}")
    else:
        launch()
    _dbg(f"conv_gemm M={m} N={nout} K={kh * kh * (c0 + c1)} stats={st is not None and st.nchunks}", out=out,
         partial=None if st is None else st.partial.reshape(-1)[:n_ * st.nchunks * ncols * 2])
    return (out, st) if stats else out


LN_FOLDED_LAUNCHES = 0   # GEMMs launched with a LayerNorm folded in (tests assert the executor took that path at the sizes it should)


class RowStats:
    """Per-row {sum, sum of squares} partials of an [M, N] tensor, written by the GEMM that produced it (conv_gemm(row_stats=...)) for the
    LayerNorm-folded GEMM that reads it (conv_gemm(ln=...)).  parts == 0: not available (the producer used another tile shape)."""

    def __init__(self, m, n, device=None):
        self.partial = empty((m, 2 * (-(-n // 320)), 2), torch.float32, device)
        self.parts = 0


def ln_rowstats_finalize(rs, c, eps):
    """RowStats (parts > 0) of an [M, c] tensor -> fp32 [M, 2] = {rstd, -mean * rstd} per row (fmx_layernorm_rowstats_finalize): the per-column
    operand of the operand-swapped LayerNorm-folded GEMM (conv_gemm(ln_swapped=...))."""
    m = rs.partial.shape[0]
    ab = empty((m, 2), torch.float32, rs.partial.device)
    _lib.check(_lib.lib().fmx_layernorm_rowstats_finalize(_p(rs.partial), rs.parts, m, c, float(eps), _p(ab), stream_ptr()),
               "fmx_layernorm_rowstats_finalize")
    return ab


def linear(x, wgt, bias=None, **kw):
    return conv_gemm(x, wgt, wgt.shape[0], bias=bias, **kw)


def geglu_interleave(w, b):
    inner = w.shape[0] // 2
    wo = torch.empty_like(w)
    bo = torch.empty_like(b) if b is not None else None
    _lib.check(_lib.lib().fmx_geglu_interleave_rows(_p(w), _p(b), _p(wo), _p(bo), inner, w.shape[1], stream_ptr()),
               "fmx_geglu_interleave_rows")
    return wo, bo


_KINDS = {torch.float16: 0, torch.float32: 1, torch.bfloat16: 2, torch.bool: 3, torch.uint8: 3}


def strided_copy4(src, dst, dims, src_strides, dst_strides):
    """dst[i0, i1, i2, i3] = convert(src[i0, i1, i2, i3]) for i < dims (element strides; base = data_ptr of each tensor): the layout /
    dtype adapter of the attention- and op-level entry points (fmx_strided_copy4).  A bool source becomes an additive mask (0 / -inf)."""
    if not (src.is_cuda and dst.is_cuda):
        raise TypeError("strided_copy4 expects device tensors")
    if src.dtype not in _KINDS or dst.dtype not in (torch.float16, torch.float32, torch.bfloat16):
        raise TypeError(f"strided_copy4: unsupported element types {src.dtype} -> {dst.dtype}")
    ss = (C.c_int64 * 4)(*[int(v) for v in src_strides])
    ds = (C.c_int64 * 4)(*[int(v) for v in dst_strides])
    dm = (C.c_int32 * 4)(*[int(v) for v in dims])
    _lib.check(_lib.lib().fmx_strided_copy4(_p(src), _KINDS[src.dtype], ss, _p(dst), _KINDS[dst.dtype], ds, dm, stream_ptr()), "fmx_strided_copy4")
    return dst


def attention(q, k, vt, *, batch, heads, nq, nk, nk_pad, dpad, scale, q_bs, q_rs, k_bs, k_rs, vt_bs, vt_hs, vt_ds,
              out=None, force32=False, causal=False, mask=None, mask_strides=(0, 0, 0)):
    """q/k/vt are base tensors (views allowed: the data_ptr is the element (0,0,0,0)); strides in elements.
    mask: optional fp16 additive mask, element (b, h, i, j) at mask_strides = (batch, head, query) element strides, rows of nk_pad keys."""
    sfx, elem = _elem(q, k, vt, out)
    fn_name = "fmx_attention" + sfx
    if out is None:
        out = empty((batch * nq, heads * dpad), elem, q.device)
    a = AttnArgs()
    a.q, a.k, a.vt, a.o = _p(q), _p(k), _p(vt), _p(out)
    a.q_bs, a.q_rs, a.k_bs, a.k_rs = q_bs, q_rs, k_bs, k_rs
    a.vt_bs, a.vt_hs, a.vt_ds = vt_bs, vt_hs, vt_ds
    a.o_bs, a.o_rs = nq * out.stride(0), out.stride(0)
    a.batch, a.heads, a.nq, a.nk, a.nk_pad, a.dpad = batch, heads, nq, nk, nk_pad, dpad
    a.scale = -float(scale) if force32 else float(scale)  # test hook: negative scale selects the 32-query-per-wave kernel
    a.zero_page = _p(zero_page(q.device))
    a.causal = 1 if causal else 0
    a.mask = _p(mask)
    a.mask_bs, a.mask_hs, a.mask_qs = (int(v) for v in mask_strides)
    if _profiler is not None:
        d_true = int(round(float(scale) ** -2))
        flops = 4.0 * batch * heads * nq * nk * d_true
        # one key tile or two (the 77-token text context): no loop over keys, the launch is one read of Q + one write of O -> its own
        # family with an HBM roofline (bench.py)
        short = nk <= 128
        _profiler.launch("attention_short_keys" if short else "attention", flops,
                         lambda: _lib.check(getattr(_lib.lib(), fn_name)(C.byref(a), stream_ptr()), fn_name),
                         tag=f"B={batch} H={heads} Nq={nq} Nk={nk} d={dpad}",
                         nbytes=2.0 * batch * heads * dpad * (2 * nq + 2 * nk) if short else 0.0)
        return out
    _lib.check(getattr(_lib.lib(), fn_name)(C.byref(a), stream_ptr()), fn_name)
    return out


def attention_single_head512(q, k, vt, out, *, batch, nq, nk, nk_pad, q_bs, q_rs, k_bs, k_rs, vt_bs, vt_ds, scale):
    """One 512-wide head, fused (fmx_attention_single_head512_f16 / _bf16): q / k token-major views, vt = V^T [512 rows][batch * nk_pad keys]."""
    sfx, _ = _elem(q, k, vt, out)

    def launch():
        _lib.check(getattr(_lib.lib(), "fmx_attention_single_head512" + sfx)(_p(q), q_bs, q_rs, _p(k), k_bs, k_rs, _p(vt), vt_bs, vt_ds, _p(out),
                                                                             nq * out.stride(0), out.stride(0), batch, nq, nk, nk_pad, float(scale),
                                                                             stream_ptr()), "fmx_attention_single_head512" + sfx)
    if _profiler is not None:
        _profiler.launch("attention", 4.0 * batch * nq * nk * 512, launch, tag=f"B={batch} H=1 Nq={nq} Nk={nk} d=512")
    else:
        launch()
    _dbg(f"attention512 B={batch} N={nq}", out=out)
    return out


def softmax_rows_(x):
    sfx, _ = _elem(x)
    _lib.check(getattr(_lib.lib(), "fmx_softmax_rows" + sfx)(_p(x), x.shape[0], x.shape[1], x.stride(0), stream_ptr()), "fmx_softmax_rows" + sfx)
    return x


def _gn_chunks(n, hw):
    """chunks per image of the stand-alone statistics pass: >= 2048 blocks on the big tensors, >= 32 pixels per chunk"""
    want = max(1, -(-2048 // n))
    return int(max(1, min(1024, want, max(1, hw // 32))))


def _stats_geometry(n, hw):
    """(chunks of the stand-alone pass, capacity of the partial buffer): the GEMM's own statistics come per 256-row tile (8-wave kernels), per 128-row
    tile (4-wave kernels, round 5) or per 512-row tile"""
    fb = _gn_chunks(n, hw)
    return fb, max(fb, hw // 128 if hw % 128 == 0 else (hw // 256 if hw % 256 == 0 else 0))


def stats_buffer(n, hw, c, device=None):
    """Workspace for the statistics of an [n, hw, c] tensor about to be produced by conv_gemm(stats=True, stats_partial=...)."""
    return empty((n, _stats_geometry(n, hw)[1], c, 2), torch.float32, device)


def attach_stats(t, st):
    """Remember the producer's statistics on the tensor OBJECT (views and copies do not inherit them; whoever writes into the tensor in
    place afterwards -- ControlNet residuals, Python hooks -- must call clear_stats).  The record carries the tensor's torch version counter
    and address: an in-place torch edit nobody announced (a hook holding the NCHW view: `h.add_(...)`) bumps the counter, and `_attached_stats`
    then ignores the stale record instead of normalising with it (C-ABI launches write through raw pointers and do not bump it: they call
    clear_stats themselves).

    The guard is BEST-EFFORT and says so: torch keeps ONE version counter per storage, and every arena tensor is a view of the
    arena's single uint8 buffer -- there the counter says nothing about THIS tensor (any `zero_()` / `copy_()` on any other arena tensor bumps it,
    which used to drop the statistics of every live tensor and silently add a statistics pass), so for views of a larger buffer, and under
    torch.inference_mode (no counters at all; the product path), the tag is the address alone and the contract is the explicit one: whoever writes
    into a tensor in place calls clear_stats.  Only tensors that own their storage (what a Python hook's own buffers are) get the version check."""
    t._fmx_gn_stats = None if st is None else (st, _version_of(t), t.data_ptr())
    return t


def _version_of(t):
    """torch's in-place version counter; inference-mode tensors (processing runs under torch.inference_mode, as the reference does) do not
    keep one -- there the address is the only tag and in-place writers have to announce themselves through clear_stats, as before."""
    if t.is_inference():
        return None
    if t.untyped_storage().nbytes() != t.numel() * t.element_size():
        return None          # a window of a larger buffer (the arena): the storage's shared counter is not about this tensor
    return t._version


def _attached_stats(t):
    rec = getattr(t, "_fmx_gn_stats", None)
    if rec is None:
        return None
    st, version, ptr = rec
    if version != _version_of(t) or ptr != t.data_ptr():
        t._fmx_gn_stats = None
        return None
    return st


def clear_stats(t):
    if t is not None and getattr(t, "_fmx_gn_stats", None) is not None:
        t._fmx_gn_stats = None
    return t


def groupnorm_stats(x):
    """x: fp16 / bf16 [N, H, W, C] / [N, HW, C] (pixel stride = x.stride(-2), channels contiguous) -> GnStats (stand-alone pass over the tensor)."""
    sfx, _ = _elem(x)
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    nch = _gn_chunks(n, hw)
    partial = empty((n, nch, c, 2), torch.float32, x.device)
    _lib.check(getattr(_lib.lib(), "fmx_groupnorm_stats" + sfx)(_p(x), c, x.stride(-2), n, hw, _p(partial), nch, stream_ptr()), "fmx_groupnorm_stats" + sfx)
    return GnStats(partial, nch)


def groupnorm(x, gamma, beta, eps, *, x1=None, silu=False, groups=32, out=None, stats=None, stats1=None):
    """x: [N, H, W, C0] (x1: [N, H, W, C1] concatenated after it) -> [N, H, W, C0+C1].  stats / stats1: GnStats of x / x1 when their
    producer left them (conv_gemm(stats=True)); a source without statistics gets its own pass here.  fp16, or bf16 throughout (the VAE's
    second element type)."""
    sfx, elem = _elem(x, x1, gamma, beta, out)
    n = x.shape[0]
    hw = x.numel() // (n * x.shape[-1])
    c0 = x.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    if out is None:
        out = empty(tuple(x.shape[:-1]) + (c0 + c1,), elem, x.device)

    if stats is None:
        stats = _attached_stats(x)
    if stats1 is None and x1 is not None:
        stats1 = _attached_stats(x1)

    def run():
        s0 = stats if stats is not None else groupnorm_stats(x)
        s1 = (stats1 if stats1 is not None else groupnorm_stats(x1)) if x1 is not None else None
        ss = empty((n, c0 + c1, 2), torch.float32, x.device)
        _lib.check(getattr(_lib.lib(), "fmx_groupnorm_apply" + sfx)(_p(x), _p(x1), c0, c1, x.stride(-2), x1.stride(-2) if x1 is not None else 0, n, hw,
                                                                    _p(s0.partial), s0.nchunks, _p(s1.partial) if s1 is not None else None,
                                                                    s1.nchunks if s1 is not None else 0, groups, float(eps), _p(gamma), _p(beta),
                                                                    1 if silu else 0, _p(ss), _p(out), stream_ptr()), "fmx_groupnorm_apply" + sfx)
    if _profiler is not None:
        have = (stats is not None) + (x1 is not None and stats1 is not None)
        _profiler.launch("groupnorm", 0.0, run, tag=f"N={n} HW={hw} C={c0}+{c1} stats_from_producer={have}/{1 + (x1 is not None)}",
                         nbytes=2.0 * 2 * n * hw * (c0 + c1))
    else:
        run()
    _dbg(f"groupnorm N={n} HW={hw} C={c0}+{c1} stats={'producer' if stats is not None else 'own'}/{'producer' if stats1 is not None else 'own'}", out=out)
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _check_f16(x, gamma, beta)
    c = x.shape[-1]
    rows = x.numel() // c
    if out is None:
        out = empty(x.shape, torch.float16, x.device)
    _lib.check(_lib.lib().fmx_layernorm_f16(_p(x), _p(gamma), _p(beta), _p(out), rows, c, float(eps), stream_ptr()), "fmx_layernorm_f16")
    return out


def rmsnorm(x, weight, eps=1e-6, out=None):
    """T5LayerNorm (backend/nn/t5.py:15-25): x * rsqrt(mean(x^2) + eps) * weight over the last dim; fp16 or bf16 throughout"""
    sfx, elem = _elem(x, weight, out)
    c = x.shape[-1]
    rows = x.numel() // c
    if out is None:
        out = empty(x.shape, elem, x.device)
    _lib.check(getattr(_lib.lib(), "fmx_rmsnorm" + sfx)(_p(x), _p(weight), _p(out), rows, c, float(eps), stream_ptr()), "fmx_rmsnorm" + sfx)
    return out


def layernorm_padded(x, gamma, beta, out, rows_per_image, eps=1e-5):
    """LayerNorm of x [B*n, C] written into out [B, n_pad, C] (rows n..n_pad of every image are left untouched)."""
    _check_f16(x, gamma, beta, out)
    c = x.shape[-1]
    rows = x.numel() // c
    assert out.is_contiguous() and out.shape[-1] == c and out.shape[0] * rows_per_image == rows and out.shape[1] >= rows_per_image
    _lib.check(_lib.lib().fmx_layernorm_padded_f16(_p(x), _p(gamma), _p(beta), _p(out), rows, c, float(eps), rows_per_image, out.shape[1],
                                                   stream_ptr()), "fmx_layernorm_padded_f16")
    return out


def layernorm_mod(x, scale, shift, rows_per_batch, eps=1e-6, out=None):
    """Flux adaLN: (1 + scale[b]) * LayerNorm(x, no affine) + shift[b]; scale/shift: [B, c] views with a common row stride."""
    sfx, elem = _elem(x, scale, shift, out)
    c = x.shape[-1]
    rows = x.numel() // c
    assert scale.stride(0) == shift.stride(0) and scale.stride(-1) == 1 and shift.stride(-1) == 1
    if out is None:
        out = empty(x.shape, elem, x.device)
    _lib.check(getattr(_lib.lib(), "fmx_layernorm_mod" + sfx)(_p(x), _p(scale), _p(shift), scale.stride(0), rows_per_batch, _p(out), rows, c,
                                                              float(eps), stream_ptr()), "fmx_layernorm_mod" + sfx)
    return out


def flux_qk_norm_rope(qkv, q_scale, k_scale, pe, q_out, k_out, vt_out, *, batch, tokens, heads, head_dim, row_off, l_pad, eps=1e-6):
    """qkv: [batch*tokens, >= 3*heads*head_dim] (row stride = qkv.stride(0)); pe: fp32 [L_total, head_dim/2, 2] (cos, sin)."""
    sfx, _ = _elem(qkv, q_scale, k_scale, q_out, k_out, vt_out)
    assert pe.dtype == torch.float32 and pe.is_contiguous()
    _lib.check(getattr(_lib.lib(), "fmx_flux_qk_norm_rope" + sfx)(_p(qkv), qkv.stride(0), _p(q_scale), _p(k_scale), _p(pe), _p(q_out), _p(k_out),
                                                                  _p(vt_out), batch, tokens, heads, head_dim, row_off, l_pad, float(eps),
                                                                  stream_ptr()), "fmx_flux_qk_norm_rope" + sfx)


def timestep_embedding(t, dim, max_period=10000.0, out=None, dtype=torch.float16):
    b = t.shape[0]
    if out is None:
        out = empty((b, dim), dtype, t.device)
    name = "fmx_timestep_embedding_bf16" if out.dtype == torch.bfloat16 else "fmx_timestep_embedding"
    _lib.check(getattr(_lib.lib(), name)(_p(t), _p(out), b, dim, float(max_period), stream_ptr()), name)
    return out


def silu(x, out=None):
    sfx, elem = _elem(x, out)
    if out is None:
        out = empty(x.shape, elem, x.device)
    _lib.check(getattr(_lib.lib(), "fmx_silu" + sfx)(_p(x), _p(out), x.numel(), stream_ptr()), "fmx_silu" + sfx)
    return out


def add_control_(h, ctrl, alpha=1.0):
    """h [B,H,W,C] fp16 NHWC += alpha * ctrl [B,C,H,W] (ControlNet residual, unet.py:44-52), in place.  A residual whose MEMORY is already
    channels-last (the native ControlNet hands out NCHW views of NHWC buffers) is added elementwise; an NCHW-contiguous one goes through
    the transposing kernel."""
    b, hh, ww, c = h.shape
    clear_stats(h)  # h changes in place: statistics its producer left are stale
    if tuple(ctrl.shape) != (b, c, hh, ww):
        raise ValueError(f"control residual {tuple(ctrl.shape)} does not match activation {(b, c, hh, ww)}")
    nhwc = ctrl.permute(0, 2, 3, 1)
    if ctrl.device == h.device and nhwc.is_contiguous() and ctrl.dtype in (torch.float16, torch.float32) and not (c == 1 or hh * ww == 1):
        _lib.check(_lib.lib().fmx_add_scaled_f16(_p(h), _p(nhwc), 1 if ctrl.dtype == torch.float32 else 0, float(alpha), h.numel(), stream_ptr()),
                   "fmx_add_scaled_f16")
        return h
    if alpha != 1.0:
        ctrl = ctrl * alpha
    ctrl = ctrl.to(device=h.device, dtype=torch.float32).contiguous()
    _lib.check(_lib.lib().fmx_add_control_nchw(_p(h), _p(ctrl), b, c, hh * ww, stream_ptr()), "fmx_add_control_nchw")
    return h


ACT_QUICK_GELU, ACT_GELU_ERF, ACT_RELU = 0, 1, 2


def avgpool2x2(x):
    """fp16 NHWC [N, H, W, C] -> [N, H/2, W/2, C]"""
    n, h, w, c = x.shape
    out = empty((n, h // 2, w // 2, c), torch.float16, x.device)
    _lib.check(_lib.lib().fmx_avgpool2x2_nhwc_f16(_p(x), _p(out), n, h, w, c, stream_ptr()), "fmx_avgpool2x2_nhwc_f16")
    return out


def act(x, kind, out=None):
    if out is None:
        out = empty(x.shape, torch.float16, x.device)
    _lib.check(_lib.lib().fmx_act_f16(_p(x), _p(out), x.numel(), int(kind), stream_ptr()), "fmx_act_f16")
    return out


def embed_tokens(ids, tok_emb, pos_emb, out=None):
    """ids int32 [B, T] -> fp16 [B*T, C] = tok_emb[ids] + pos_emb[t]"""
    b, t = ids.shape
    c = tok_emb.shape[1]
    if out is None:
        out = empty((b * t, c), torch.float16, ids.device)
    _lib.check(_lib.lib().fmx_embed_tokens(_p(ids), _p(tok_emb), _p(pos_emb), _p(out), b, t, c, tok_emb.shape[0], stream_ptr()), "fmx_embed_tokens")
    return out


def cast_f16(x, out=None):
    if out is None:
        out = empty(x.shape, torch.float16, x.device)
    _lib.check(_lib.lib().fmx_cast_f32_to_f16(_p(x), _p(out), x.numel(), stream_ptr()), "fmx_cast_f32_to_f16")
    return out


def unet_pack_input(x, sigma, reps, sigma_data=1.0, out=None):
    b, c, h, w = x.shape
    if out is None:
        out = empty((reps * b * h * w, 64), torch.float16, x.device)
    _lib.check(_lib.lib().fmx_unet_pack_input(_p(x), _p(sigma), float(sigma_data), b, c, h, w, reps, _p(out), stream_ptr()),
               "fmx_unet_pack_input")
    return out


def im2col3x3_smallc(x, c, out=None):
    n, h, w, ld = x.shape
    if out is None:
        out = empty((n * h * w, 64), x.dtype, x.device)    # a 16-bit word shuffle: fp16 and bf16 alike
    _lib.check(_lib.lib().fmx_im2col3x3_smallc(_p(x), ld, n, c, h, w, _p(out), stream_ptr()), "fmx_im2col3x3_smallc")
    return out


PREDICTION_TYPES = {"epsilon": 0, "const": 0, "v_prediction": 1, "edm": 2}


def cfg_combine(eps, ld_eps, x, sigma, reps, cond_scale, denoised=None, cond_pred=None, uncond_pred=None, prediction_type="epsilon",
                sigma_data=1.0):
    b, c, h, w = x.shape
    if denoised is None:
        denoised = torch.empty_like(x)
    _lib.check(_lib.lib().fmx_cfg_combine(_p(eps), ld_eps, _p(x), _p(sigma), b, c, h, w, reps, float(cond_scale), _p(denoised),
                                          _p(cond_pred), _p(uncond_pred), PREDICTION_TYPES[prediction_type], float(sigma_data), stream_ptr()),
               "fmx_cfg_combine")
    return denoised


def euler_step(x, denoised, sigma, sigma_next, noise=None, noise_scale=0.0, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().fmx_sampler_euler_step(_p(x), _p(denoised), float(sigma), float(sigma_next), _p(noise), float(noise_scale),
                                                 _p(out), x.numel(), stream_ptr()), "fmx_sampler_euler_step")
    return out


def lincomb3(x, d0, d1, a, b, c, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().fmx_sampler_lincomb3(_p(x), _p(d0), _p(d1), float(a), float(b), float(c), _p(out), x.numel(), stream_ptr()),
               "fmx_sampler_lincomb3")
    return out


def lincomb(srcs, coefs, out=None):
    """out = sum_k coefs[k] * srcs[k] over fp32 tensors of one shape; one fused pass per 8 terms."""
    import ctypes
    n = len(srcs)
    assert n >= 1 and n == len(coefs)
    if n > 8:  # more terms than one launch takes (UniPC at order > 5): accumulate in groups, the partial sum riding along as term 0
        acc = lincomb(srcs[:8], coefs[:8], out=out)
        for i in range(8, n, 7):
            acc = lincomb([acc] + list(srcs[i:i + 7]), [1.0] + list(coefs[i:i + 7]), out=acc)
        return acc
    x = srcs[0]
    for t in srcs:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == x.shape and t.device == x.device
    if out is None:
        out = torch.empty_like(x)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    cf = (ctypes.c_float * n)(*[float(c) for c in coefs])
    _lib.check(_lib.lib().fmx_sampler_lincomb(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cf, ctypes.c_void_p), n, _p(out), x.numel(),
                                              stream_ptr()), "fmx_sampler_lincomb")
    return out


def error_norm(x_low, x_high, x_prev, atol, rtol):
    """-> python float: ||(x_low - x_high) / max(atol, rtol * max(|x_low|, |x_prev|))|| / sqrt(numel) (one device sync: the adaptive
    step-size controller needs the value on the host to accept / reject the step)."""
    ws = torch.empty(257, dtype=torch.float32, device=x_low.device)
    _lib.check(_lib.lib().fmx_sampler_error_norm(_p(x_low), _p(x_high), _p(x_prev), float(atol), float(rtol), _p(ws), ws[256:].data_ptr(),
                                                 x_low.numel(), stream_ptr()), "fmx_sampler_error_norm")
    return float(ws[256])


def resize_separable(x, ystart, yweights, xstart, xweights):
    """x fp32 [..., H, W] -> [..., OH, OW] with per-axis (start, weights) tables (host or device tensors), see fmx.h."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    h, w = x.shape[-2:]
    oh, ky = yweights.shape
    ow, kx = xweights.shape
    dev = x.device
    ys, yw = ystart.to(device=dev, dtype=torch.int32).contiguous(), yweights.to(device=dev, dtype=torch.float32).contiguous()
    xs, xw = xstart.to(device=dev, dtype=torch.int32).contiguous(), xweights.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty(x.shape[:-2] + (oh, ow), dtype=torch.float32, device=dev)
    planes = x.numel() // (h * w)
    _lib.check(_lib.lib().fmx_resize_separable_f32(_p(x), _p(out), _p(ys), _p(yw), _p(xs), _p(xw), planes, h, w, oh, ow, ky, kx, stream_ptr()),
               "fmx_resize_separable_f32")
    return out


def scale_f32(x, s, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().fmx_scale_f32(_p(x), float(s), _p(out), x.numel(), stream_ptr()), "fmx_scale_f32")
    return out


def blend_masked(a, a_mask, b, b_mask, out=None):
    """out = a * a_mask + b * b_mask (fp32, same shapes); out may alias a or b."""
    for t in (a, a_mask, b, b_mask):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.shape != a.shape:
            raise TypeError("blend_masked expects contiguous fp32 device tensors of one shape")
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.lib().fmx_blend_masked(_p(a), _p(a_mask), _p(b), _p(b_mask), _p(out), a.numel(), stream_ptr()), "fmx_blend_masked")
    return out


def _vae_sfx(dtype):
    return "_bf16" if dtype == torch.bfloat16 else ""


def vae_pack_latent(z, scaling_factor, shift, ld=8, out=None, dtype=torch.float16):
    b, c, h, w = z.shape
    if out is None:
        out = empty((b, h, w, ld), dtype, z.device)
    name = "fmx_vae_pack_latent" + _vae_sfx(out.dtype)
    _lib.check(getattr(_lib.lib(), name)(_p(z), float(scaling_factor), float(shift), b, c, h, w, _p(out), ld, stream_ptr()), name)
    return out


def vae_sample_posterior(moments, ld, noise, lc, scale=1.0, shift=0.0, out=None):
    """moments fp16 [B*npix, ld] (mean | logvar), noise fp32 [B, lc, h, w] -> fp32 [B, lc, h, w] = (sample - shift) * scale."""
    b, _, hh, ww = noise.shape
    if out is None:
        out = torch.empty_like(noise)
    name = "fmx_vae_sample_posterior" + _vae_sfx(moments.dtype)
    _lib.check(getattr(_lib.lib(), name)(_p(moments), ld, _p(noise), b, lc, hh * ww, float(scale), float(shift), _p(out), stream_ptr()), name)
    return out


def count_nonfinite(x):
    """-> python int: inf / NaN values in the fp16 tensor x (contiguous).  One device sync -- the caller decides on the host (VAE overflow guard)."""
    _check_f16(x)
    assert x.is_contiguous()
    cnt = torch.empty(1, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().fmx_count_nonfinite_f16(_p(x), x.numel(), _p(cnt), stream_ptr()), "fmx_count_nonfinite_f16")
    return int(cnt.item())


def conv3x3_narrow(x, wgt, bias, nout, out=None, ld_out=4):
    """3x3 convolution (stride 1, padding 1) with at most 4 output channels, direct kernel (fmx_conv3x3_narrow): x NHWC [n, h, w, c] with c a
    multiple of 32, wgt [nout, 9 * c] in the GEMM's tap-major layout -> [n * h * w, ld_out] (columns >= nout zero when ld_out == 4)."""
    sfx, elem = _elem(x, wgt, bias)
    n, h, w, c = x.shape
    if out is None:
        out = empty((n * h * w, ld_out), elem, x.device)
    name = "fmx_conv3x3_narrow" + sfx

    def run():
        _lib.check(getattr(_lib.lib(), name)(_p(x), n, h, w, c, _p(wgt), _p(bias), nout, _p(out), ld_out, stream_ptr()), name)
    if _profiler is not None:   # HBM-bound: one read of the input, one write of the [npix, ld_out] output
        _profiler.launch("conv3x3_narrow", 2.0 * n * h * w * 9 * c * nout, run, tag=f"N={n} H={h} W={w} C={c} nout={nout}", nbytes=2.0 * n * h * w * (c + ld_out))
    else:
        run()
    return out


# A/B knob: FMX_UP2X=0 keeps the Upsample convolutions on the implicit GEMM with nearest-upsample-on-load (9 taps per output pixel, round 4's path)
_UP2X = _lib.knob("FMX_UP2X", "1") != "0"


def fold_up2x_weights(wk, c):
    """[nout, 9 * c] (the GEMM layout: taps (ky, kx) row-major, then channels) -> [4, nout, 4 * c]: the tap sums of the four parity phases of
    conv3x3(nearest_upsample_x2(x)) -- phase 2 * py + px, its 2 x 2 taps (dy, dx) row-major (include/fmx.h fmx_conv3x3_up2x).  Even output rows
    see input rows {iy - 1, iy} through {w[0], w[1] + w[2]}, odd ones {iy, iy + 1} through {w[0] + w[1], w[2]}; columns alike.  Summed in fp32,
    rounded ONCE to the element type."""
    nout = wk.shape[0]
    w = wk.reshape(nout, 3, 3, c).float()

    def fold(t, dim, parity):
        a, b, d = t.unbind(dim)
        return torch.stack([a, b + d] if parity == 0 else [a + b, d], dim)
    phases = [fold(fold(w, 1, py), 2, px).reshape(nout, 4 * c) for py in (0, 1) for px in (0, 1)]
    return torch.stack(phases).to(wk.dtype).contiguous()


def conv3x3_up2x_supported(x, nout, out_hw=None):
    """the four-phase form of an Upsample convolution (fmx_conv3x3_up2x) takes: an exact x2 output, input channels in 64s, an input width in 32s,
    a whole number of 256-pixel statistics chunks per image, an input within one launch's 32-bit offsets"""
    if x.dim() != 4 or not x.is_contiguous():
        return False
    n, h, w, c = x.shape
    if out_hw is not None and tuple(out_hw) != (2 * h, 2 * w):
        return False
    return c % 64 == 0 and w % 32 == 0 and nout % 8 == 0 and (h * w) % 256 == 0 and x.numel() * 2 < 3.0e9


def conv3x3_up2x_eligible(x, nout, out_hw=None):
    """... and where the executors USE it: launches that fill the chip.  A phase is a launch of its own on 256-row tiles (n h w / 256 x ceil(nout / 320) of
    them); the nine-tap form is ONE launch with four times the rows.  Measured (profiles/r50_small_batch_breakdowns.jsonl): 32 tiles per phase (SDXL at
    UNet batch 2) 0.51 ms = 210 TFLOP/s against ~0.40 ms for the nine taps; 64 tiles per phase (SD1.5 at UNet batch 8) level; 256 / 512 tiles
    (SDXL at UNet batch 16) 0.65 + 0.70 ms against 1.48 + 1.39.  From 128 tiles per phase on."""
    if not _UP2X or not conv3x3_up2x_supported(x, nout, out_hw):
        return False
    n, h, w, _ = x.shape
    return (n * h * w // 256) * -(-nout // 320) >= 128


def conv3x3_up2x(x, w4, bias, nout, *, stats=True, stats_partial=None):
    """conv3x3(pad 1) of the x2 nearest-upsampled x as four 2 x 2 convolutions on x's own grid (backend/nn/unet.py:340-355, backend/nn/vae.py:42-57):
    x NHWC [n, h, w, c], w4 from fold_up2x_weights -> (out [n * 2h * 2w, nout], GnStats of out | None)."""
    sfx, elem = _elem(x, w4, bias)
    n, h, w, c = x.shape
    assert w4.shape == (4, nout, 4 * c) and w4.is_contiguous()
    out = empty((n * 4 * h * w, nout), elem, x.device)
    st = None
    partial, cap = None, 0
    if stats and _FUSED_STATS:
        cap = _stats_geometry(n, 4 * h * w)[1]
        partial = stats_partial if stats_partial is not None else empty((n, cap, nout, 2), torch.float32, x.device)
        cap = partial.numel() // (n * nout * 2)
        st = GnStats(partial, 0)
    nch = C.c_int32(0)
    name = "fmx_conv3x3_up2x" + sfx

    def run():
        _lib.check(getattr(_lib.lib(), name)(_p(x), n, h, w, c, _p(w4), _p(bias), nout, _p(out), _p(partial), cap, C.byref(nch), _p(zero_page(x.device)),
                                            stream_ptr()), name)
        if st is not None:
            st.nchunks = nch.value
    if _profiler is not None:   # the multiply-adds EXECUTED: 4 taps per output pixel (the reference's form of the layer has 9)
        _profiler.launch("gemm_conv", 2.0 * n * 4 * h * w * nout * 4 * c, run, tag=f"M={n * 4 * h * w} N={nout} K={4 * c} kh=2 s=1 up2x-phases{' +gnstats' if st is not None else ''}")
    else:
        run()
    return (out, st) if stats else out


# A/B knob: FMX_CONV_GN_FUSE=0 keeps GroupNorm apply + implicit-GEMM convolution as two launches where conv3x3_gn_silu is eligible (round 5's path)
_CONV_GN_FUSE = _lib.knob("FMX_CONV_GN_FUSE", "1") != "0"


def conv3x3_gn_silu_eligible(x, cout, stats=None):
    """the fused GroupNorm + SiLU + 3x3 convolution kernel covers: 128 output channels, input channels a multiple of 64, statistics of x available from
    its producer, a tensor large enough that the 8 x 32-pixel tiles fill the chip (the VAE decoder's last level); everything else takes
    groupnorm() + conv_gemm()"""
    if not _CONV_GN_FUSE or x.dim() != 4 or cout != 128 or x.shape[-1] % 64 or x.shape[-1] > 1024 or not x.is_contiguous():
        return False
    n, h, w, _ = x.shape
    if (stats if stats is not None else _attached_stats(x)) is None or n * h * w < (1 << 19):
        return False
    tiles = -(-h // 8) * -(-w // 32)
    return tiles <= _stats_geometry(n, h * w)[1]


def conv3x3_gn_silu(x, gamma, beta, eps, wgt, bias, *, residual=None, out=None, groups=32, stats=None, want_stats=True, stats_partial=None):
    """silu(group_norm(x)) -> conv3x3(stride 1, pad 1) -> + bias (+ residual), ONE launch (fmx_conv3x3_gn_silu, csrc/fmx_conv_patch.hip;
    backend/nn/vae.py:98-114).  x NHWC [n, h, w, cin] with its producer's GnStats (`stats`, or attached to x); wgt [128, 9 * cin];
    -> (out [n*h*w, 128], GnStats of out | None).  Check conv3x3_gn_silu_eligible first."""
    sfx, elem = _elem(x, gamma, beta, wgt, bias, residual)
    n, h, w, cin = x.shape
    cout = wgt.shape[0]
    s0 = stats if stats is not None else _attached_stats(x)
    assert s0 is not None and wgt.shape[1] == 9 * cin and wgt.is_contiguous()
    if out is None:
        out = empty((n * h * w, cout), elem, x.device)
    ss = empty((n, cin, 2), torch.float32, x.device)
    st = None
    a = ConvGnArgs()
    a.x, a.n, a.h, a.w, a.cin = _p(x), n, h, w, cin
    a.x_partial, a.x_nchunks, a.groups, a.eps = _p(s0.partial), s0.nchunks, groups, float(eps)
    a.gamma, a.beta, a.scale_shift = _p(gamma), _p(beta), _p(ss)
    a.wgt, a.cout, a.bias = _p(wgt), cout, _p(bias)
    a.residual, a.ld_res = _p(residual), (residual.stride(-2) if residual is not None else 0)
    a.out, a.ld_out = _p(out), out.stride(-2)
    nch = C.c_int32(0)
    if want_stats and _FUSED_STATS:
        cap = _stats_geometry(n, h * w)[1]
        partial = stats_partial if stats_partial is not None else empty((n, cap, cout, 2), torch.float32, x.device)
        a.stats, a.stats_cap = _p(partial), partial.numel() // (n * cout * 2)
        st = GnStats(partial, 0)
    name = "fmx_conv3x3_gn_silu" + sfx

    def run():
        _lib.check(getattr(_lib.lib(), name)(C.byref(a), C.byref(nch), stream_ptr()), name)
        if st is not None:
            st.nchunks = nch.value
    if _profiler is not None:
        _profiler.launch("gemm_conv", 2.0 * n * h * w * cout * 9 * cin, run, tag=f"M={n * h * w} N={cout} K={9 * cin} kh=3 s=1 gn+silu fused{' +gnstats' if st is not None else ''}")
    else:
        run()
    return out, st


def conv3x3_narrow_gn_silu(x, gamma, beta, eps, wgt, bias, nout, *, groups=32, out=None, ld_out=4, stats=None):
    """conv3x3_narrow on silu(group_norm(x)) without storing it (fmx_conv3x3_narrow_gn_silu): the norm_out -> swish -> conv_out tail of the VAE decoder.  x NHWC with its
    producer's GnStats (`stats`, or attached to x); FMX_CONV_GN_FUSE=0 (or no statistics) -> the caller keeps groupnorm() + conv3x3_narrow()."""
    sfx, elem = _elem(x, gamma, beta, wgt, bias)
    n, h, w, c = x.shape
    s0 = stats if stats is not None else _attached_stats(x)
    assert s0 is not None
    if out is None:
        out = empty((n * h * w, ld_out), elem, x.device)
    ss = empty((n, c, 2), torch.float32, x.device)
    name = "fmx_conv3x3_narrow_gn_silu" + sfx

    def run():
        _lib.check(getattr(_lib.lib(), name)(_p(x), n, h, w, c, _p(s0.partial), s0.nchunks, groups, float(eps), _p(gamma), _p(beta), _p(ss), _p(wgt), _p(bias), nout,
                                            _p(out), ld_out, stream_ptr()), name)
    if _profiler is not None:   # HBM-bound: one read of the input, one write of the [npix, ld_out] output
        _profiler.launch("conv3x3_narrow", 2.0 * n * h * w * 9 * c * nout, run, tag=f"N={n} H={h} W={w} C={c} nout={nout} gn+silu fused", nbytes=2.0 * n * h * w * (c + ld_out))
    else:
        run()
    return out


def vae_unpack_image(y, ld, npix, c, out):
    name = "fmx_vae_unpack_image" + _vae_sfx(y.dtype)
    _lib.check(getattr(_lib.lib(), name)(_p(y), ld, npix, c, _p(out), stream_ptr()), name)
    return out


def philox_randn(seed, offset, n, device, want_raw=False):
    out = torch.empty(n, dtype=torch.float32, device=device)
    raw = torch.empty((n, 4), dtype=torch.int32, device=device) if want_raw else None
    _lib.check(_lib.lib().fmx_philox_randn(C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(offset)), _p(out), _p(raw), n, stream_ptr()),
               "fmx_philox_randn")
    return (out, raw) if want_raw else out
