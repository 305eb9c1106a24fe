"""MI355X-native executor for Forge's LDM UNet (SD1.x / SD2.x / SDXL family).

Drop-in for `IntegratedUNet2DConditionModel` (reference: backend/nn/unet.py:481-763): same class name, same
`forward(x, timesteps, context, y, control, transformer_options)` contract and the same LDM checkpoint keys,
but no nn.Module graph underneath -- a flat layout (layout.py) walked once per step, every op a hand-written
gfx950 kernel reached through the C-ABI (include/fmx.h).

Differences from the reference that are design, not omissions:
  * activations are fp16 NHWC end to end, so `b c h w -> b (h w) c` (unet.py:315,324) costs nothing;
  * GroupNorm+SiLU is one op and reads the un-materialised concat [h ; skip] directly (unet.py:741);
  * bias, ResBlock time-embedding add (unet.py:469-477), residual adds (:240,:272,:478), GEGLU (:104-111) and the
    nearest Upsample (:340-355) are epilogue / loader options of the MFMA GEMM, not separate kernels;
  * to_q|to_k run as one GEMM, V is produced already transposed (V^T = Wv X^T, operand swap) for the fused
    attention kernel; head dims that are not MFMA-tile multiples (SD1.5: 40) are zero-padded in the weights;
  * cross-attention K / V^T of the text context and the SDXL label embedding depend only on the conditioning,
    so they are computed once per job and cached (`prepare_context`), not once per step (unet.py:145-155, :707);
  * all ResBlock `emb_layers` projections run as ONE GEMM per step.
ControlNet residuals (`control={'input': [...], 'middle': [...], 'output': [...]}`, unet.py:44-52,714,732,739) are injected
natively (fp32 NCHW residual -> fp16 NHWC activation, one transposing add kernel each); such forwards run eagerly, not from
the captured graph, since the residuals change every step.  Hooks that need per-block Python callbacks on NCHW tensors
(`patches`, `patches_replace`, `block_modifiers`) are called eagerly on [B, N, C] / NCHW views; the module-typed hooks
(`block_inner_modifiers`, `group_norm_wrapper`) get stand-ins for the modules (below).
"""

import os

import itertools

import torch

from ... import hipops as ops
from ..._lib import knob as _knob
from ...runtime import Arena, ArenaOverflow
from .layout import ConvIn, Down, Res, SpatialT, Up, unet_layout

SUPPORTED_DPAD = (48, 64, 80, 160)
CTX_PAD = 64  # text tokens are padded to a multiple of the attention key tile


# ---- what the two module-typed hooks get to see (unet.py:73-91 `block_inner_modifiers`, :436-474 / :755-757 `group_norm_wrapper`) ----------------
# The reference hands them torch.nn.Module objects.  The executor has no modules, so it hands over light stand-ins that carry the reference's
# CLASS NAMES and the attributes extensions look at (type checks by isinstance against this module's classes or by type(layer).__name__, channel
# counts, the GroupNorm's parameters), and -- for the GroupNorm -- are callable: norm(x) runs the native kernel.
class _Layer:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Conv2d(_Layer):
    pass


class ResBlock(_Layer):
    pass


class SpatialTransformer(_Layer):
    pass


class Upsample(_Layer):
    pass


class Downsample(_Layer):
    pass


class TimestepEmbedSequential(list):
    """the block a layer belongs to, as `block_inner_modifiers` receive it: a sequence of the stand-ins above"""


class GroupNorm(_Layer):
    """stand-in for the ResBlock's / output head's GroupNorm32: num_groups, num_channels, eps, weight, bias; norm(x) -> group_norm(x)"""

    def __call__(self, x):
        xh = x.permute(0, 2, 3, 1).to(torch.float16).contiguous()
        y = ops.groupnorm(xh, self.weight, self.bias, self.eps, silu=False, groups=self.num_groups)
        return y.permute(0, 3, 1, 2).to(x.dtype)

    forward = __call__


_LN_FOLD = _knob("FMX_LN_FOLD", "1") != "0"   # A/B knob: 0 keeps the LayerNorm kernels in front of attn2.to_q / ff.net.0
_LN_FOLD1 = _knob("FMX_LN_FOLD1", "1") != "0"  # A/B knob: 0 keeps norm1 as a kernel (round 2) while norm2 / norm3 stay folded
# A/B knob (round 5): 1 runs attn2 as ONE launch where the fused epilogue is eligible (csrc/fmx_gemm256p.hip XA).  Default 0: built, correct, and measured
# slower (SDXL batch 8: 102.4 -> 104.9 ms per step, profiles/r32_cross_attention_in_to_q_epilogue_ab.jsonl) -- the 77-key attention is vector-issue work,
# not a memory pass the fusion deletes
_XATTN_FUSE = _knob("FMX_XATTN_FUSE", "0") == "1"


def _fold_layernorm(wt, bias, gamma, beta):
    """LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (W beta + b)  with  W' = W * gamma  (per input channel).  -> (W' fp16, colsum fp32 of
    the fp16 values of W' -- it has to cancel exactly what the MFMAs accumulate -- and the folded bias fp16).  unet.py:262-279."""
    wf = (wt.float() * gamma.float()[None, :]).to(torch.float16).contiguous()
    colsum = wf.float().sum(1).contiguous()
    b2 = wt.float() @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return wf, colsum, b2.to(torch.float16).contiguous()


def _dpad(d):
    # heads up to 64 wide are padded to 64, not to the next MFMA granule (SD1.5's d = 40 would fit 48): the 64-wide attention kernel is the
    # tuned one -- 1030 vs 438 TFLOP/s at 4096 tokens outweighs the 33 % of zero columns (SD1.5 512^2 batch 4: 15.8 -> 14.6 ms/step, profiles/r06g_*)
    if d <= 64:
        return 64
    for s in SUPPORTED_DPAD:
        if d <= s:
            return s
    raise NotImplementedError(f"head dim {d} > 160 not supported by the fused attention kernel")


def _conv_w(w):
    """torch conv weight [Cout, Cin, kh, kw] -> GEMM weight [Cout, kh*kw*Cin] (K ordered ky, kx, c)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def _pad_head_rows(w, heads, d, dp):
    if d == dp:
        return w
    k = w.shape[1]
    out = w.new_zeros(heads, dp, k)
    out[:, :d] = w.reshape(heads, d, k)
    return out.reshape(heads * dp, k)


def _pad_head_cols(w, heads, d, dp):
    if d == dp:
        return w
    n = w.shape[0]
    out = w.new_zeros(n, heads, dp)
    out[:, :, :d] = w.reshape(n, heads, d)
    return out.reshape(n, heads * dp)


class ContextCache:
    """Per-conditioning tensors reused by every step: padded context, cross-attention K / V^T per transformer block,
    label-embedding MLP output."""

    def __init__(self):
        self.key = None
        self.bu = 0
        self.tokens = 0
        self.tpad = 0
        self.kv = {}       # block key -> (K [Bu*Tp, H*dp], Vt [H*dp, Bu*Tp])
        self.label = None  # [Bu, time_embed_dim] or None


_EXEC_SERIAL = itertools.count(1)   # process-wide: an executor's serial is never handed out twice (id() of a freed one can be)


class IntegratedUNet2DConditionModel:
    encoder_only = False  # cnets/cldm.py's ControlNet re-uses this executor for its trunk (input blocks + middle block)
    TRUNK_PREFIXES = ("input_blocks.", "middle_block.", "time_embed.", "label_emb.")
    RETAIN_TRUNK_WEIGHTS = True

    def __init__(self, config, state_dict, device="cuda", arena_bytes=None):
        self.config = dict(config)
        self.layout = unet_layout(config, self.encoder_only)
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.storage_dtype = self.computation_dtype = torch.float16
        self.in_channels = self.layout.in_channels
        self.model_channels = self.layout.model_channels
        self.out_channels = self.layout.out_channels
        self.num_classes = config.get("num_classes")
        self._pad_bufs = {}  # (Bu, n_pad, C) -> persistent zero-padded LayerNorm output for ragged token counts (see _attn_block)
        self._arena = None
        self.exec_serial = next(_EXEC_SERIAL)  # identity of this executor in graph keys (k_model.py): unlike id(), never re-used
        self.arena_epoch = 0  # bumped whenever the arena is re-allocated: graphs captured on the old one hold dangling pointers
        self._arena_bytes = arena_bytes
        self._ctx = ContextCache()
        # {transformer block key: (norm1, norm2, norm3 ran folded into the projections behind them)} of the LAST forward that walked the layers
        # (eager or graph capture) -- which rounding sites exist depends on it, so the executor-faithful oracle (oracle/unet_fp16sites.py, tests
        # only) is told what happened instead of guessing the dispatcher's tile choices
        self.fold_trace = {}
        # Upsample layers whose convolution ran as four phase convolutions on tap-summed weights in the last forward (another rounding site of the
        # weights: the rounding oracle is told, like fold_trace)
        self.up2x_trace = set()
        # tap(name, tensor): when set, called on the eager path with every layer's stored output (name = the layer's LDM key; `<SpatialTransformer
        # key>.proj_in`, every `...transformer_blocks.N` and its `.attn1` / `.attn2` (the stream after those sub-layers), `.attn1.o` / `.attn2.o`
        # (attention outputs, heads at their padded width), `.attn{1,2}.{q,k,v}` (the attention kernel's operands), `.ff.g` (GEGLU output), a ResBlock's `.h` (conv1 + emb), "time_embed", "out.2") -- a VIEW of the kernel's own buffer, to be copied by the callee.
        # The layer-wise parity tests feed these tensors to the rounding oracle layer by layer (tests/test_gpu_sharp_parity.py).
        self.tap = None
        # Control-LoRA builds its control model from the UNet's own trunk weights (patcher/controlnet.py:445-453 reads
        # `diffusion_model.state_dict()`); the kernel layouts below are not invertible in general (fused / padded / transposed), so the source
        # tensors of the trunk are kept by reference (no copy) under their LDM keys
        # (RETAIN_TRUNK_WEIGHTS = False drops them: a throughput run that never attaches a Control-LoRA saves the duplicate)
        self._trunk_sd = {k: v for k, v in state_dict.items() if k.startswith(self.TRUNK_PREFIXES)} if self.RETAIN_TRUNK_WEIGHTS else None
        self._load(state_dict)

    # ------------------------------------------------------------------------------------------------------------
    # weights: LDM keys -> resident fp16 tensors in kernel layouts
    # ------------------------------------------------------------------------------------------------------------
    def _load(self, sd):
        dev = self.device
        lay = self.layout

        def T(key):
            return sd[key].to(device=dev, dtype=torch.float16).contiguous()

        def lin(key, bias=True):
            return T(key + ".weight"), (T(key + ".bias") if bias and (key + ".bias") in sd else None)

        w = {}
        w["te0"], w["te2"] = lin("time_embed.0"), lin("time_embed.2")
        if lay.adm_in_channels is not None:
            w["le0"], w["le2"] = lin("label_emb.0.0"), lin("label_emb.0.2")
        emb_w, emb_b, emb_off = [], [], {}
        off = 0
        for L in lay.all_layers():
            k = L.key
            if isinstance(L, ConvIn):
                full = sd[k + ".weight"].to(dev, torch.float16)            # [mc, cin, 3, 3]
                lc = min(lay.out_channels, L.cin) if L.cin * 9 > 64 else L.cin
                # inpainting / edit models (in_channels 9 / 8 = latent + c_concat, k_model.py:38-39): the conv is linear in its input
                # channels, so it is split into the per-step part on the noisy latent and a per-JOB part on the concat conditioning
                # (prepare_concat), which rides into the first GEMM as its residual operand
                parts = []
                for lo, hi in ((0, lc), (lc, L.cin)):
                    if hi > lo:
                        cw = _conv_w(full[:, lo:hi].contiguous())               # [mc, 9*(hi-lo)]
                        if cw.shape[1] > 64:
                            raise NotImplementedError("each half of the first conv must have 9*channels <= 64 (im2col'ed)")
                        wp = cw.new_zeros(cw.shape[0], 64)
                        wp[:, :cw.shape[1]] = cw
                        parts.append(wp.contiguous())
                w[k] = (parts[0], T(k + ".bias"))
                self.latent_channels, self.concat_channels = lc, L.cin - lc
                if len(parts) > 1:
                    w[k + ".concat"] = parts[1]
            elif isinstance(L, Res):
                w[k + ".gn1"] = (T(k + ".in_layers.0.weight"), T(k + ".in_layers.0.bias"))
                w[k + ".conv1"] = (_conv_w(sd[k + ".in_layers.2.weight"].to(dev, torch.float16)), T(k + ".in_layers.2.bias"))
                w[k + ".gn2"] = (T(k + ".out_layers.0.weight"), T(k + ".out_layers.0.bias"))
                w[k + ".conv2"] = (_conv_w(sd[k + ".out_layers.3.weight"].to(dev, torch.float16)), T(k + ".out_layers.3.bias"))
                if L.has_skip_conv:
                    w[k + ".skip"] = (_conv_w(sd[k + ".skip_connection.weight"].to(dev, torch.float16)), T(k + ".skip_connection.bias"))
                emb_w.append(T(k + ".emb_layers.1.weight"))
                emb_b.append(T(k + ".emb_layers.1.bias"))
                emb_off[k] = (off, L.cout)
                off += L.cout
            elif isinstance(L, SpatialT):
                H, d = L.heads, L.dim_head
                dp = _dpad(d)
                inner = H * d
                if (H * dp) % 64 != 0:
                    raise NotImplementedError(f"heads*padded_dim = {H}*{dp} must be a multiple of 64 (GEMM K tile)")
                w[k + ".norm"] = (T(k + ".norm.weight"), T(k + ".norm.bias"))
                pin_w = sd[k + ".proj_in.weight"].to(dev, torch.float16).reshape(inner, L.ch).contiguous()
                pout_w = sd[k + ".proj_out.weight"].to(dev, torch.float16).reshape(L.ch, inner).contiguous()
                w[k + ".proj_in"] = (pin_w, T(k + ".proj_in.bias"))
                w[k + ".proj_out"] = (pout_w, T(k + ".proj_out.bias"))
                for di in range(L.depth):
                    b = f"{k}.transformer_blocks.{di}"
                    for nname in ("norm1", "norm2", "norm3"):
                        w[f"{b}.{nname}"] = (T(f"{b}.{nname}.weight"), T(f"{b}.{nname}.bias"))
                    q1 = _pad_head_rows(T(b + ".attn1.to_q.weight"), H, d, dp)
                    k1 = _pad_head_rows(T(b + ".attn1.to_k.weight"), H, d, dp)
                    w[b + ".attn1.qk"] = torch.cat([q1, k1], 0).contiguous()
                    w[b + ".attn1.v"] = _pad_head_rows(T(b + ".attn1.to_v.weight"), H, d, dp).contiguous()
                    w[b + ".attn1.out"] = (_pad_head_cols(T(b + ".attn1.to_out.0.weight"), H, d, dp).contiguous(), T(b + ".attn1.to_out.0.bias"))
                    w[b + ".attn2.q"] = _pad_head_rows(T(b + ".attn2.to_q.weight"), H, d, dp).contiguous()
                    w[b + ".attn2.k"] = _pad_head_rows(T(b + ".attn2.to_k.weight"), H, d, dp).contiguous()
                    w[b + ".attn2.v"] = _pad_head_rows(T(b + ".attn2.to_v.weight"), H, d, dp).contiguous()
                    w[b + ".attn2.out"] = (_pad_head_cols(T(b + ".attn2.to_out.0.weight"), H, d, dp).contiguous(), T(b + ".attn2.to_out.0.bias"))
                    w[b + ".ff1"] = ops.geglu_interleave(T(b + ".ff.net.0.proj.weight"), T(b + ".ff.net.0.proj.bias"))
                    w[b + ".ff2"] = lin(b + ".ff.net.2")
                    if _LN_FOLD and L.ch % 320 == 0:
                        # norm2 / norm3 folded into the projection behind them (ops.conv_gemm(ln=...)): weights scaled by gamma, their fp32
                        # column sums, beta pushed through the weight into the bias.  Kept beside the plain weights: small problems (where
                        # the producing GEMM runs on another tile shape and emits no row statistics) and hooked runs use the LayerNorm kernel.
                        w[b + ".attn2.q.ln"] = _fold_layernorm(w[b + ".attn2.q"], None, *w[b + ".norm2"])
                        w[b + ".ff1.ln"] = _fold_layernorm(*w[b + ".ff1"], *w[b + ".norm3"])
                        if (H * dp) % 320 == 0:
                            # norm1 (round 3): into the q|k projection the same way, and into the operand-swapped V^T projection, where the
                            # LayerNorm rows are the output's columns (fmx.h ln_col_ab / ln_row_cb): the weight is the row operand there
                            g1, b1 = w[b + ".norm1"]
                            w[b + ".attn1.qk.ln"] = _fold_layernorm(w[b + ".attn1.qk"], None, g1, b1)
                            wv = w[b + ".attn1.v"]
                            wvf = (wv.float() * g1.float()[None, :]).to(torch.float16).contiguous()
                            w[b + ".attn1.v.ln"] = (wvf, torch.stack([wvf.float().sum(1), wv.float() @ b1.float()], 1).contiguous())
            elif isinstance(L, Down):
                w[k] = (_conv_w(sd[k + ".op.weight"].to(dev, torch.float16)), T(k + ".op.bias"))
            elif isinstance(L, Up):
                w[k] = (_conv_w(sd[k + ".conv.weight"].to(dev, torch.float16)), T(k + ".conv.bias"))
                # the same layer as four 2 x 2 convolutions on the un-upsampled grid (4 / 9 of the multiply-adds, hipops.conv3x3_up2x); the 3 x 3 form
                # stays for output sizes that are not an exact x2 (unet.py:349-351 `output_shape`) and for hooked runs
                w[k + ".up2x"] = ops.fold_up2x_weights(w[k][0], w[k][0].shape[1] // 9)
        w["emb_all"] = (torch.cat(emb_w, 0).contiguous(), torch.cat(emb_b, 0).contiguous())
        self._emb_off = emb_off
        self._emb_total = off
        if not self.encoder_only:
            w["out.gn"] = (T("out.0.weight"), T("out.0.bias"))
            w["out.conv"] = (_conv_w(sd["out.2.weight"].to(dev, torch.float16)), T("out.2.bias"))
        self.w = w
        self._load_extra(sd, w)
        torch.cuda.synchronize(dev)

    def _load_extra(self, sd, w):
        pass

    # ------------------------------------------------------------------------------------------------------------
    # conditioning-only work (once per job)
    # ------------------------------------------------------------------------------------------------------------
    def prepare_context(self, context, y=None):
        """context [Bu, T, Dc] (any float dtype, device), y [Bu, adm] or None.  Cached on tensor identity+version."""
        def ver(t):
            try:
                return t._version
            except RuntimeError:  # inference-mode tensors carry no version counter
                return -1
        key = (context.data_ptr(), tuple(context.shape), ver(context),
               None if y is None else (y.data_ptr(), tuple(y.shape), ver(y)))
        c = self._ctx
        if c.key == key:
            return c
        bu, t, dc = context.shape
        tp = -(-t // CTX_PAD) * CTX_PAD
        ctx = torch.zeros(bu, tp, dc, dtype=torch.float16, device=self.device)
        ctx[:, :t] = context.to(device=self.device, dtype=torch.float16)
        ctx2d = ctx.reshape(bu * tp, dc)
        c.kv = {}
        for L in self.layout.all_layers():
            if not isinstance(L, SpatialT):
                continue
            for di in range(L.depth):
                b = f"{L.key}.transformer_blocks.{di}"
                kc = torch.empty(bu * tp, self.w[b + ".attn2.k"].shape[0], dtype=torch.float16, device=self.device)
                ops.linear(ctx2d, self.w[b + ".attn2.k"], out=kc, ld_out=kc.shape[1])
                vt = torch.empty(self.w[b + ".attn2.v"].shape[0], bu * tp, dtype=torch.float16, device=self.device)
                ops.conv_gemm(self.w[b + ".attn2.v"], ctx2d, bu * tp, out=vt, ld_out=bu * tp)
                c.kv[b] = (kc, vt)
        c.label = None
        if self.layout.adm_in_channels is not None:
            assert y is not None and y.shape[0] == bu, "SDXL-style UNet needs y (unet.py:702,706)"
            yh = y.to(device=self.device, dtype=torch.float16).contiguous()
            h1 = ops.linear(yh, *self.w["le0"], out=torch.empty(bu, self.layout.time_embed_dim, dtype=torch.float16, device=self.device))
            h1 = ops.silu(h1, out=h1)
            c.label = ops.linear(h1, *self.w["le2"], out=torch.empty(bu, self.layout.time_embed_dim, dtype=torch.float16, device=self.device))
        c.key, c.bu, c.tokens, c.tpad = key, bu, t, tp
        # the cached K / V^T / label tensors above are NEW allocations: a captured graph that still points at the previous ones must not be
        # replayed even if `key` repeats (a later conditioning tensor can land on the address of an earlier, freed one) -- KModel compares this
        c.serial = getattr(c, "serial", 0) + 1
        c.ctx = ctx[:, :t]  # fp16 [Bu, T, Dc]: the `context` a transformer hook sees
        self._ctx_keepalive = (context, y)
        return c

    # ------------------------------------------------------------------------------------------------------------
    # layers (NHWC fp16 [Bu, H, W, C])
    # ------------------------------------------------------------------------------------------------------------
    def _res(self, L, x, skip, emb_all, arena):
        """ResBlock (unet.py:433-478).  The GroupNorm statistics of every tensor a convolution writes here come out of that GEMM's
        epilogue (ops.conv_gemm(stats=True)) and travel with the tensor object (ops.attach_stats): no GroupNorm re-reads its input for them."""
        k = L.key
        bu, hh, ww, _ = x.shape
        out = ops.empty((bu, hh, ww, L.cout))
        out_part = ops.stats_buffer(bu, hh * ww, L.cout)
        m = arena.mark()
        g1 = ops.groupnorm(x, *self.w[k + ".gn1"], 1e-5, x1=skip, silu=True)
        off, cout = self._emb_off[k]
        h, h_st = ops.conv_gemm(g1, self.w[k + ".conv1"][0], cout, kh=3, pad=1, bias=self.w[k + ".conv1"][1],
                                rowvec=emb_all[:, off:off + cout], stats=True)
        self._tap(k + ".h", h.view(bu, hh, ww, cout))
        g2 = ops.groupnorm(h.view(bu, hh, ww, cout), *self.w[k + ".gn2"], 1e-5, silu=True, stats=h_st)
        if L.has_skip_conv:
            sk = ops.conv_gemm(x, self.w[k + ".skip"][0], cout, x1=skip, bias=self.w[k + ".skip"][1])
        else:
            sk = x.view(-1, cout)
        _, st = ops.conv_gemm(g2, self.w[k + ".conv2"][0], cout, kh=3, pad=1, bias=self.w[k + ".conv2"][1], residual=sk,
                              out=out.view(-1, cout), ld_out=cout, stats=True, stats_partial=out_part)
        arena.release(m)
        return ops.attach_stats(out, st)

    def _attn_block(self, b, L, h, bu, n, ctxc, arena, rs1=None, want_next=False):
        """One BasicTransformerBlock on h [M, C] (updated in place).  rs1: the row statistics the projection that WROTE h left (proj_in or the
        previous block's ff.net.2) -- norm1 then runs folded into the q|k and V^T projections; want_next: leave the statistics of this block's
        output for the next block's norm1.  -> that RowStats (or None)."""
        H, d = L.heads, L.dim_head
        dp = _dpad(d)
        hd = H * dp
        m_tok = bu * n
        # LayerNorm folding (norm2 -> attn2.to_q, norm3 -> ff.net.0): the projection that writes h also leaves per-row sums of what it wrote
        fold = (b + ".ff1.ln") in self.w
        rs2 = ops.RowStats(m_tok, h.shape[1]) if fold else None
        rs3 = ops.RowStats(m_tok, h.shape[1]) if fold else None
        rs_next = ops.RowStats(m_tok, h.shape[1]) if (want_next and fold) else None
        mk = arena.mark()
        # self attention
        folded1 = n % 64 == 0 and rs1 is not None and bool(rs1.parts) and (b + ".attn1.v.ln") in self.w
        if folded1:
            wqk, csqk, bqk = self.w[b + ".attn1.qk.ln"]
            ab = ops.empty((m_tok, 2), torch.float32)     # {rstd, -mean rstd} per token: the q|k GEMM derives them anyway and leaves them for V^T
            qk = ops.conv_gemm(h, wqk, wqk.shape[0], bias=bqk, ln=(rs1, csqk, 1e-5), ln_ab_out=ab)   # [M, 2*H*dp] = [Q | K] of LN(h)
            wvf, vcb = self.w[b + ".attn1.v.ln"]
            vt = ops.conv_gemm(wvf, h, m_tok, ln_swapped=(ab, vcb))                                   # [H*dp, M] = V^T of LN(h)
            self._tap_qkv(b + ".attn1", qk[:, :hd].view(bu, n, hd), qk[:, hd:].view(bu, n, hd), vt.view(hd, bu, n))
            o = ops.attention(qk, qk[:, hd:], vt, batch=bu, heads=H, nq=n, nk=n, nk_pad=n, dpad=dp, scale=d ** -0.5,
                              q_bs=n * 2 * hd, q_rs=2 * hd, k_bs=n * 2 * hd, k_rs=2 * hd, vt_bs=n, vt_hs=dp * m_tok, vt_ds=m_tok)
        elif n % 64 == 0:
            n1 = ops.layernorm(h, *self.w[b + ".norm1"])
            qk = ops.linear(n1, self.w[b + ".attn1.qk"])                   # [M, 2*H*dp] = [Q | K]
            vt = ops.conv_gemm(self.w[b + ".attn1.v"], n1, m_tok)          # [H*dp, M] = V^T (operand swap)
            self._tap_qkv(b + ".attn1", qk[:, :hd].view(bu, n, hd), qk[:, hd:].view(bu, n, hd), vt.view(hd, bu, n))
            o = ops.attention(qk, qk[:, hd:], vt, batch=bu, heads=H, nq=n, nk=n, nk_pad=n, dpad=dp, scale=d ** -0.5,
                              q_bs=n * 2 * hd, q_rs=2 * hd, k_bs=n * 2 * hd, k_rs=2 * hd, vt_bs=n, vt_hs=dp * m_tok, vt_ds=m_tok)
        else:
            # ragged token count (latent H*W not a multiple of the 64-key tile, e.g. SDXL at 832x1216): LayerNorm writes each image's tokens
            # at a padded stride into a persistent zero-filled buffer, so Q|K and V^T are still ONE batched GEMM each over [Bu * n_pad]
            # rows; the pad rows are zeros in, zeros out (no bias on q / k / v) and masked by nk in the attention kernel
            npad = -(-n // 64) * 64
            key = (bu, npad, h.shape[1])
            padbuf = self._pad_bufs.get(key)
            if padbuf is None:
                if len(self._pad_bufs) >= 24:   # bounded; captured graphs that point at the dropped buffers are invalidated via the epoch
                    torch.cuda.synchronize(self.device)
                    self._pad_bufs.clear()
                    self.arena_epoch += 1
                padbuf = self._pad_bufs[key] = torch.zeros(bu, npad, h.shape[1], dtype=torch.float16, device=self.device)
            ops.layernorm_padded(h, *self.w[b + ".norm1"], out=padbuf, rows_per_image=n)
            p2d = padbuf.view(bu * npad, -1)
            qk = ops.linear(p2d, self.w[b + ".attn1.qk"])                  # [Bu*n_pad, 2*H*dp]
            vt = ops.conv_gemm(self.w[b + ".attn1.v"], p2d, bu * npad)     # [H*dp, Bu*n_pad]
            o = ops.attention(qk, qk[:, hd:], vt, batch=bu, heads=H, nq=n, nk=n, nk_pad=npad, dpad=dp, scale=d ** -0.5,
                              q_bs=npad * 2 * hd, q_rs=2 * hd, k_bs=npad * 2 * hd, k_rs=2 * hd, vt_bs=npad,
                              vt_hs=dp * bu * npad, vt_ds=bu * npad)
        self._tap(b + ".attn1.o", o.view(bu, n, -1))
        ops.linear(o, *self.w[b + ".attn1.out"], residual=h, out=h, ld_out=h.shape[1], row_stats=rs2)
        arena.release(mk)
        self._tap(b + ".attn1", h.view(bu, n, -1))
        self.fold_trace[b] = (folded1, bool(fold and rs2.parts), False)
        # cross attention against the cached text K / V^T
        kc, vtc = ctxc.kv[b]
        tp = ctxc.tpad
        # (round 5, off by default -- see _XATTN_FUSE) the whole of attn2 in ONE launch: a 256 x 320 tile of the query projection holds five whole 64-wide heads
        # of 256 queries of one image, so the attention against the cached K / V^T runs in that GEMM's epilogue out of the accumulators (fmx.h xa_*)
        fused2 = (_XATTN_FUSE and fold and bool(rs2.parts) and self.tap is None and d == 64 and hd % 320 == 0 and n % 256 == 0 and ctxc.tokens <= 80 and tp >= 80)
        if fused2:
            wq, csq, bq = self.w[b + ".attn2.q.ln"]
            o2 = ops.conv_gemm(h, wq, wq.shape[0], bias=bq, ln=(rs2, csq, 1e-5), xattn=(kc, vtc, ctxc.tokens, tp, n, d ** -0.5))
        else:
            if fold and rs2.parts:
                wq, csq, bq = self.w[b + ".attn2.q.ln"]
                q2 = ops.conv_gemm(h, wq, wq.shape[0], bias=bq, ln=(rs2, csq, 1e-5))
            else:
                n2 = ops.layernorm(h, *self.w[b + ".norm2"])
                q2 = ops.linear(n2, self.w[b + ".attn2.q"])
            self._tap_qkv(b + ".attn2", q2.view(bu, n, hd), kc.view(bu, tp, hd)[:, :ctxc.tokens], vtc.view(hd, bu, tp)[:, :, :ctxc.tokens])
            o2 = ops.attention(q2, kc, vtc, batch=bu, heads=H, nq=n, nk=ctxc.tokens, nk_pad=tp, dpad=dp, scale=d ** -0.5,
                               q_bs=n * hd, q_rs=hd, k_bs=tp * hd, k_rs=hd, vt_bs=tp, vt_hs=dp * bu * tp, vt_ds=bu * tp)
        self._tap(b + ".attn2.o", o2.view(bu, n, -1))
        ops.linear(o2, *self.w[b + ".attn2.out"], residual=h, out=h, ld_out=h.shape[1], row_stats=rs3)
        arena.release(mk)
        self._tap(b + ".attn2", h.view(bu, n, -1))
        # GEGLU feed-forward
        self.fold_trace[b] = self.fold_trace[b][:2] + (bool(fold and rs3.parts),)
        if fold and rs3.parts:
            fw, csf, fb = self.w[b + ".ff1.ln"]
            g = ops.conv_gemm(h, fw, fw.shape[0], bias=fb, act=ops.ACT_GEGLU, ln=(rs3, csf, 1e-5))
        else:
            n3 = ops.layernorm(h, *self.w[b + ".norm3"])
            fw, fb = self.w[b + ".ff1"]
            g = ops.conv_gemm(n3, fw, fw.shape[0], bias=fb, act=ops.ACT_GEGLU)
        self._tap(b + ".ff.g", g.view(bu, n, -1))
        ops.linear(g, *self.w[b + ".ff2"], residual=h, out=h, ld_out=h.shape[1], row_stats=rs_next)
        arena.release(mk)
        return rs_next

    # ---- per-block Python hooks (unet.py:186-279 `patches` / `patches_replace`): eager, general-shape path ----------------------------
    def _attend(self, wq, wk, wv, H, d, xq, xk, xv):
        """Attention with arbitrary query / key / value sources (what attn{1,2}_patch may hand back): xq [B, Nq, Cq], xk / xv [B, Nk, Ck]
        fp16 -> [B*Nq, H*dp].  Keys / values are projected per image into zero-padded 64-key tiles, as the ragged self-attention path does."""
        dp = _dpad(d)
        hd = H * dp
        B, nq = xq.shape[0], xq.shape[1]
        nk = xk.shape[1]
        nkp = -(-nk // 64) * 64
        q = ops.linear(xq.reshape(B * nq, -1).to(torch.float16).contiguous(), wq)
        kbuf = torch.zeros(B, nkp, hd, dtype=torch.float16, device=self.device)
        vt = torch.zeros(hd, B * nkp, dtype=torch.float16, device=self.device)
        for bi in range(B):
            ops.linear(xk[bi].to(torch.float16).contiguous(), wk, out=kbuf[bi, :nk], ld_out=hd)
            ops.conv_gemm(wv, xv[bi].to(torch.float16).contiguous(), nk, out=vt[:, bi * nkp:bi * nkp + nk], ld_out=B * nkp)
        return ops.attention(q, kbuf, vt, batch=B, heads=H, nq=nq, nk=nk, nk_pad=nkp, dpad=dp, scale=d ** -0.5, q_bs=nq * hd, q_rs=hd,
                             k_bs=nkp * hd, k_rs=hd, vt_bs=nkp, vt_hs=dp * B * nkp, vt_ds=B * nkp)

    @staticmethod
    def _unpad_heads(t2d, B, n, H, d, dp):
        """[B*n, H*dp] (heads padded to the MFMA-friendly width) -> [B, n, H*d], the layout a replace-hook expects."""
        return t2d.view(B, n, H, dp)[..., :d].reshape(B, n, H * d)

    @staticmethod
    def _pad_heads(t, H, d, dp):
        B, n, _ = t.shape
        out = torch.zeros(B, n, H, dp, dtype=torch.float16, device=t.device)
        out[..., :d] = t.reshape(B, n, H, d).to(torch.float16)
        return out.view(B * n, H * dp)

    def _attn_block_hooked(self, b, L, h, bu, n, ctxc, arena, to):
        """BasicTransformerBlock._forward (unet.py:183-279) with its hook points, on [Bu, N, C] fp16 tensors (h is updated in place)."""
        H, d = L.heads, L.dim_head
        dp = _dpad(d)
        hd = H * dp
        C_ = h.shape[1]
        patches, replace = to.get("patches", {}), to.get("patches_replace", {})
        extra = {k: v for k, v in to.items() if k not in ("patches", "patches_replace")}
        extra["n_heads"], extra["dim_head"] = H, d
        block, block_index = to.get("block", None), to.get("block_index", 0)
        transformer_block = (block[0], block[1], block_index) if block is not None else None
        wqk = self.w[b + ".attn1.qk"]

        def sublayer(which, nrm, ctx_default, wq, wk, wv, wout):
            context, value = ctx_default, None
            if which + "_patch" in patches:
                if context is None:
                    context = nrm
                value = context
                for p in patches[which + "_patch"]:
                    nrm, context, value = p(nrm, context, value, extra)
            rep = replace.get(which, {})
            key = transformer_block if transformer_block in rep else block
            if context is None:
                context = nrm
            if value is None:
                value = context
            if key in rep:
                B_, nq, nk = nrm.shape[0], nrm.shape[1], context.shape[1]
                q = self._unpad_heads(ops.linear(nrm.reshape(B_ * nq, -1).to(torch.float16).contiguous(), wq), B_, nq, H, d, dp)
                k = self._unpad_heads(ops.linear(context.reshape(B_ * nk, -1).to(torch.float16).contiguous(), wk), B_, nk, H, d, dp)
                # the V weight is stored for the operand-swapped V^T GEMM; the same rows serve as a plain Linear weight
                v = self._unpad_heads(ops.linear(value.reshape(B_ * nk, -1).to(torch.float16).contiguous(), wv), B_, nk, H, d, dp)
                o = self._pad_heads(rep[key](q, k, v, extra), H, d, dp)
            else:
                o = self._attend(wq, wk, wv, H, d, nrm, context, value)
            out = ops.linear(o, *wout).view(bu, -1, C_)
            for p in patches.get(which + "_output_patch", []):
                out = p(out, extra)
            return out

        hv = h.view(bu, n, C_)
        n1 = ops.layernorm(h, *self.w[b + ".norm1"]).view(bu, n, C_)
        hv += sublayer("attn1", n1, None, wqk[:hd], wqk[hd:], self.w[b + ".attn1.v"], self.w[b + ".attn1.out"]).to(torch.float16)
        for p in patches.get("middle_patch", []):
            r = p(hv, extra)
            if r is not hv:
                hv.copy_(r)
        n2 = ops.layernorm(h, *self.w[b + ".norm2"]).view(bu, n, C_)
        hv += sublayer("attn2", n2, ctxc.ctx, self.w[b + ".attn2.q"], self.w[b + ".attn2.k"], self.w[b + ".attn2.v"],
                       self.w[b + ".attn2.out"]).to(torch.float16)
        n3 = ops.layernorm(h, *self.w[b + ".norm3"])
        fw, fb = self.w[b + ".ff1"]
        g = ops.conv_gemm(n3, fw, fw.shape[0], bias=fb, act=ops.ACT_GEGLU)
        ops.linear(g, *self.w[b + ".ff2"], residual=h, out=h, ld_out=h.shape[1])

    @staticmethod
    def _call_nchw(fn, h, *args):
        """Run a UNet-level hook that expects NCHW on an NHWC fp16 activation: it gets a permuted VIEW (in-place edits land in h);
        a new tensor coming back is converted to NHWC fp16."""
        v = h.permute(0, 3, 1, 2)
        ops.clear_stats(h)  # the hook may edit the view in place
        r = fn(v, *args)
        if r is v or (r.data_ptr() == h.data_ptr() and r.shape == v.shape and r.stride() == v.stride()):
            return h
        return r.permute(0, 2, 3, 1).to(torch.float16).contiguous()

    def _spatial_transformer(self, L, x, ctxc, arena, to=None):
        k = L.key
        bu, hh, ww, c = x.shape
        n = hh * ww
        inner = L.heads * L.dim_head
        out = ops.empty((bu, hh, ww, c))
        out_part = ops.stats_buffer(bu, n, c)
        mk = arena.mark()
        g = ops.groupnorm(x, *self.w[k + ".norm"], 1e-6)
        hooked = to is not None and (to.get("patches") or to.get("patches_replace"))
        # norm1 of every block folded into its q|k / V^T projections: the GEMM that writes h (proj_in, then each block's ff.net.2) leaves the row sums
        fold1 = _LN_FOLD and _LN_FOLD1 and not hooked and n % 64 == 0 and (f"{k}.transformer_blocks.0.attn1.v.ln") in self.w
        rs1 = ops.RowStats(bu * n, inner) if fold1 else None
        h = ops.linear(g.view(-1, c), *self.w[k + ".proj_in"], row_stats=rs1)  # 1x1 conv == Linear in NHWC
        self._tap(k + ".proj_in", h.view(bu, n, -1))
        for di in range(L.depth):
            if hooked:
                to["block_index"] = di
                self._attn_block_hooked(f"{k}.transformer_blocks.{di}", L, h, bu, n, ctxc, arena, to)
            else:
                rs1 = self._attn_block(f"{k}.transformer_blocks.{di}", L, h, bu, n, ctxc, arena, rs1=rs1, want_next=fold1 and di + 1 < L.depth)
            self._tap(f"{k}.transformer_blocks.{di}", h.view(bu, n, -1))
        _, st = ops.linear(h, *self.w[k + ".proj_out"], residual=x.view(-1, c), out=out.view(-1, c), ld_out=c, n=bu, h=hh, w=ww, stats=True,
                           stats_partial=out_part)
        arena.release(mk)
        return ops.attach_stats(out, st)

    def _tap(self, name, t):
        if self.tap is not None:
            self.tap(name, t)

    def _tap_qkv(self, name, q, k, vt):
        """attention operands as the kernel reads them: q / k [B, N, H*dp], V^T [H*dp, B, Nk] (handed on as [B, Nk, H*dp] views)"""
        if self.tap is not None:
            self.tap(name + ".q", q)
            self.tap(name + ".k", k)
            self.tap(name + ".v", vt.permute(1, 2, 0))

    def _layer_standin(self, L):
        if isinstance(L, Res):
            return ResBlock(channels=L.cin, out_channels=L.cout, key=L.key)
        if isinstance(L, SpatialT):
            return SpatialTransformer(in_channels=L.ch, n_heads=L.heads, d_head=L.dim_head, depth=L.depth, key=L.key)
        if isinstance(L, Down):
            return Downsample(key=L.key)
        if isinstance(L, Up):
            return Upsample(key=L.key)
        return Conv2d(key=L.key)

    def _run_block(self, blk, h, skip, emb_all, ctxc, arena, up_to=None, to=None):
        inner = to.get("block_inner_modifiers", []) if to is not None else []
        wrapper = to.get("group_norm_wrapper") if to is not None else None
        standins = TimestepEmbedSequential(self._layer_standin(L) for L in blk) if inner else None
        for li, L in enumerate(blk):
            if inner:
                # unet.py:77-79.  The first layer of an output block sees the concatenated [h, skip] tensor, as the reference's does (:741)
                if skip is not None:
                    cat = torch.cat([h, skip], dim=-1)
                    for m in inner:
                        cat = self._call_nchw(m, cat, "before", standins[li], li, standins, to)
                    h, skip = cat[..., :h.shape[-1]].contiguous(), cat[..., h.shape[-1]:].contiguous()
                else:
                    for m in inner:
                        h = self._call_nchw(m, h, "before", standins[li], li, standins, to)
            if isinstance(L, Res) and wrapper is not None:
                h = self._res_wrapped(L, h, skip, emb_all, wrapper, to)
                skip = None
            elif isinstance(L, Res):
                h = self._res(L, h, skip, emb_all, arena)
                skip = None
            elif isinstance(L, SpatialT):
                h = self._spatial_transformer(L, h, ctxc, arena, to)
                if to is not None and "transformer_index" in to:
                    to["transformer_index"] += 1  # unet.py:82-84
            elif isinstance(L, Down):
                bu, hh, ww, c = h.shape
                h, st = ops.conv_gemm(h, self.w[L.key][0], c, kh=3, stride=2, pad=1, bias=self.w[L.key][1], stats=True)
                h = ops.attach_stats(h.view(bu, (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1, c), st)
            elif isinstance(L, Up):
                bu, hh, ww, c = h.shape
                uh, uw = up_to if up_to is not None else (hh * 2, ww * 2)
                if ops.conv3x3_up2x_eligible(h, c, (uh, uw)):
                    h, st = ops.conv3x3_up2x(h, self.w[L.key + ".up2x"], self.w[L.key][1], c)
                    self.up2x_trace.add(L.key)
                else:
                    h, st = ops.conv_gemm(h, self.w[L.key][0], c, kh=3, pad=1, up=(uh, uw), bias=self.w[L.key][1], stats=True)
                h = ops.attach_stats(h.view(bu, uh, uw, c), st)
            else:
                raise TypeError(L)
            for m in inner:
                h = self._call_nchw(m, h, "after", standins[li], li, standins, to)   # unet.py:90-91
            self._tap(L.key, h)
        return h

    def _wrapped_norm(self, wrapper, key, x, eps, to):
        """group_norm_wrapper(norm, x, transformer_options) -> h (unet.py:436-474): x as an NCHW view of the channels-last activation, the
        GroupNorm as a callable stand-in; what comes back goes through SiLU (the rest of in_layers / out_layers) on the device."""
        gamma, beta = self.w[key]
        norm = GroupNorm(num_groups=32, num_channels=gamma.numel(), eps=eps, weight=gamma, bias=beta, affine=True)
        r = wrapper(norm, x.permute(0, 3, 1, 2), to)
        h = r.permute(0, 2, 3, 1).to(torch.float16).contiguous()
        # a wrapper that bypasses the norm hands x itself back (no copy above): SiLU out of place, x is still the ResBlock's skip input
        return ops.silu(h) if h.data_ptr() == x.data_ptr() else ops.silu(h, out=h)

    def _res_wrapped(self, L, x, skip, emb_all, wrapper, to):
        """ResBlock with a group_norm_wrapper installed (eager, general path): both GroupNorms go through the wrapper."""
        k = L.key
        bu, hh, ww, _ = x.shape
        xin = x if skip is None else torch.cat([x, skip], dim=-1)
        g1 = self._wrapped_norm(wrapper, k + ".gn1", xin, 1e-5, to)
        off, cout = self._emb_off[k]
        h = ops.conv_gemm(g1, self.w[k + ".conv1"][0], cout, kh=3, pad=1, bias=self.w[k + ".conv1"][1], rowvec=emb_all[:, off:off + cout])
        g2 = self._wrapped_norm(wrapper, k + ".gn2", h.view(bu, hh, ww, cout), 1e-5, to)
        sk = ops.conv_gemm(xin, self.w[k + ".skip"][0], cout, bias=self.w[k + ".skip"][1]) if L.has_skip_conv else xin.reshape(-1, cout)
        out = ops.conv_gemm(g2, self.w[k + ".conv2"][0], cout, kh=3, pad=1, bias=self.w[k + ".conv2"][1], residual=sk)
        return out.view(bu, hh, ww, cout)

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _apply_control(h, control, name):
        """unet.py:44-52: pop the last residual of control[name]; None entries are skipped."""
        if control is not None and name in control and len(control[name]) > 0:
            ctrl = control[name].pop()
            if ctrl is not None:
                ops.add_control_(h, ctrl)
        return h

    def prepare_concat(self, c_concat, bu):
        """c_concat [B or Bu, concat_channels, h, w] (mask + masked-image latent of an inpainting model; NOT input-scaled, k_model.py:38-39)
        -> conv_in's contribution of those channels, fp16 [Bu*h*w, model_channels].  Cached on the tensor's identity: once per job."""
        key = (c_concat.data_ptr(), tuple(c_concat.shape), bu)
        hit = getattr(self, "_concat_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        if self.concat_channels == 0 or c_concat.shape[1] != self.concat_channels:
            raise ValueError(f"this UNet takes {self.concat_channels} concat channels, got {tuple(c_concat.shape)}")
        cc = c_concat.to(device=self.device, dtype=torch.float32)
        if cc.shape[0] != bu:
            cc = cc.repeat(bu // cc.shape[0], 1, 1, 1)
        xl = ops.vae_pack_latent(cc.contiguous(), 1.0, 0.0, ld=8)        # NCHW fp32 -> NHWC fp16 (8-channel rows)
        term = ops.linear(ops.im2col3x3_smallc(xl, self.concat_channels), self.w[self.layout.input_blocks[0][0].key + ".concat"]).clone()
        self._concat_cache = (key, term, c_concat)
        return term

    def _forward_impl(self, xcol, t, ctxc, bu, hh, ww, arena, control=None, to=None, concat_term=None):
        """xcol: [Bu*H*W, 64] im2col of the (scaled) input; t: [Bu] fp32 table indices.  -> eps [Bu*H*W, out_ch].
        `to`: transformer_options with Python hooks (unet.py:696-763 hook points), or None on the fast path."""
        if control is not None:
            control = {k: list(v) for k, v in control.items()}
        lay = self.layout
        patches = to.get("patches", {}) if to is not None else {}
        modifiers = to.get("block_modifiers", []) if to is not None else []
        inner = to.get("block_inner_modifiers", []) if to is not None else []
        if to is not None:
            to["original_shape"] = [bu, lay.in_channels, hh, ww]
            to["transformer_index"] = 0

        def modify(h, when):
            for m in modifiers:
                h = self._call_nchw(m, h, when, to)
            return h
        t_emb = ops.timestep_embedding(t, lay.model_channels)
        e1 = ops.linear(t_emb, *self.w["te0"])
        e1 = ops.silu(e1, out=e1)
        emb = ops.linear(e1, *self.w["te2"], residual=ctxc.label)        # + label_emb(y) (unet.py:707)
        self._tap("time_embed", emb)
        se = ops.silu(emb)
        emb_all = ops.linear(se, *self.w["emb_all"])                      # every ResBlock's emb_layers at once
        hs = []
        h = None
        for bi, blk in enumerate(lay.input_blocks):
            if to is not None:
                to["block"] = ("input", bi)
            if bi == 0:
                if modifiers or inner:
                    # the 'before' hook of block 0 sees the network input: the centre tap of the im2col rows (already fp16); re-packed after
                    if self.concat_channels:
                        raise NotImplementedError("block modifiers on the input of an inpainting UNet")
                    ci = lay.in_channels
                    x_mod = modify(xcol.view(bu, hh, ww, -1)[..., 4 * ci:5 * ci].contiguous(), "before")
                    conv_in = TimestepEmbedSequential([Conv2d(key=blk[0].key, in_channels=ci, out_channels=lay.model_channels)])
                    for m in inner:
                        x_mod = self._call_nchw(m, x_mod, "before", conv_in[0], 0, conv_in, to)
                    xcol = ops.unet_pack_input(x_mod.permute(0, 3, 1, 2).float().contiguous(), torch.zeros(bu, dtype=torch.float32, device=self.device),
                                               1, 1.0)
                cw, cb = self.w[blk[0].key]
                if (concat_term is None) != (self.concat_channels == 0):
                    raise ValueError("an inpainting / edit UNet needs c_concat (and only such a UNet takes one)")
                h, st = ops.linear(xcol, cw, cb, residual=concat_term, n=bu, h=hh, w=ww, stats=True)
                h = ops.attach_stats(h.view(bu, hh, ww, lay.model_channels), st)
                self._tap(blk[0].key, h)
                for m in inner:
                    h = self._call_nchw(m, h, "after", conv_in[0], 0, conv_in, to)
            else:
                h = modify(h, "before")
                h = self._run_block(blk, h, None, emb_all, ctxc, arena, to=to)
            h = self._apply_control(h, control, "input")
            h = modify(h, "after")
            for p in patches.get("input_block_patch", []):
                h = self._call_nchw(p, h, to)
            hs.append(h)
            for p in patches.get("input_block_patch_after_skip", []):
                h = self._call_nchw(p, h, to)
        if to is not None:
            to["block"] = ("middle", 0)
        h = modify(h, "before")
        h = self._run_block(lay.middle, h, None, emb_all, ctxc, arena, to=to)
        h = self._apply_control(h, control, "middle")
        h = modify(h, "after")
        for bi, blk in enumerate(lay.output_blocks):
            if to is not None:
                to["block"] = ("output", bi)
            skip = self._apply_control(hs.pop(), control, "output")
            for p in patches.get("output_block_patch", []):
                hv, sv = h.permute(0, 3, 1, 2), skip.permute(0, 3, 1, 2)
                ops.clear_stats(h)
                ops.clear_stats(skip)
                rh, rs = p(hv, sv, to)
                h = h if rh is hv else rh.permute(0, 2, 3, 1).to(torch.float16).contiguous()
                skip = skip if rs is sv else rs.permute(0, 2, 3, 1).to(torch.float16).contiguous()
            up_to = (hs[-1].shape[1], hs[-1].shape[2]) if hs else None
            if modifiers:
                # the reference hands block modifiers the CONCATENATED [h, skip] tensor (unet.py:741-748); the executor keeps the two
                # halves separate (dual-input GroupNorm / conv), so it is materialised only when a modifier wants to see it
                cat = modify(torch.cat([h, skip], dim=-1), "before")
                h, skip = cat[..., :h.shape[-1]].contiguous(), cat[..., h.shape[-1]:].contiguous()
            h = self._run_block(blk, h, skip, emb_all, ctxc, arena, up_to, to=to)
            h = modify(h, "after")
        if to is not None:
            to["block"] = ("last", 0)
        h = modify(h, "before")
        oc = lay.out_channels
        wrapped = to is not None and to.get("group_norm_wrapper") is not None
        if (not wrapped and oc <= 4 and h.dim() == 4 and h.is_contiguous() and h.shape[-1] % 32 == 0 and ops._CONV_GN_FUSE and ops._attached_stats(h) is not None
                and h.shape[0] * h.shape[1] * h.shape[2] >= (1 << 16)):
            # out.0 -> SiLU -> out.2 (unet.py:755-764) in one launch: the normalisation applied while the direct kernel stages its patch (round 6)
            out = ops.conv3x3_narrow_gn_silu(h, *self.w["out.gn"], 1e-5, self.w["out.conv"][0], self.w["out.conv"][1], oc, ld_out=oc)
            self._tap("out.2", out.view(bu, hh, ww, -1)[..., :oc])
            if modifiers:
                out = modify(out.view(bu, hh, ww, -1)[..., :oc].contiguous(), "after").reshape(bu * hh * ww, oc)
            return out
        if wrapped:
            g = self._wrapped_norm(to["group_norm_wrapper"], "out.gn", h, 1e-5, to)       # unet.py:755-758
        else:
            g = ops.groupnorm(h, *self.w["out.gn"], 1e-5, silu=True)
        cg = g.shape[-1]
        if oc <= 4 and cg % 32 == 0 and g.dim() == 4 and g.is_contiguous() and g.numel() * 2 < 3.0e9:
            # the `out` convolution (320 -> 4 channels, unet.py:760-764) as the direct narrow-output kernel (round 4): the implicit GEMM spent 0.21 ms per step on it
            out = ops.conv3x3_narrow(g, self.w["out.conv"][0], self.w["out.conv"][1], oc, ld_out=oc)
        else:
            out = ops.conv_gemm(g, self.w["out.conv"][0], oc, kh=3, pad=1, bias=self.w["out.conv"][1])
        self._tap("out.2", out.view(bu, hh, ww, -1)[..., :oc])
        if modifiers:
            out = modify(out.view(bu, hh, ww, -1)[..., :oc].contiguous(), "after").reshape(bu * hh * ww, oc)
        return out

    def _get_arena(self, bu, hh, ww):
        need = self._arena_bytes or max(1 << 28, 40 * bu * hh * ww * self.layout.model_channels * 2)
        if self._arena is None or self._arena.capacity < need:
            torch.cuda.synchronize(self.device)  # nothing may still be running out of the arena that is about to be freed
            self._arena = None
            self._arena = Arena(need, self.device)
            self.arena_epoch += 1
        return self._arena

    def forward_packed(self, xcol, t, ctxc, bu, hh, ww, control=None, transformer_options=None, concat_term=None):
        """Hot-path entry (no layout conversion): returns eps as fp16 [Bu*H*W, out_channels] living in the arena
        (valid until the next forward)."""
        while True:
            arena = self._get_arena(bu, hh, ww)
            arena.reset()
            try:
                with arena:
                    return self._forward_impl(xcol, t, ctxc, bu, hh, ww, arena, control, self._hooks(transformer_options), concat_term)
            except ArenaOverflow:
                torch.cuda.synchronize(self.device)
                self._arena_bytes = arena.capacity * 2
                self._arena = None
                self.arena_epoch += 1

    def state_dict(self):
        """LDM-keyed source tensors of the encoder trunk (the part a ControlNet shares with the UNet); see __init__."""
        if self._trunk_sd is None:
            raise RuntimeError("this executor was built with RETAIN_TRUNK_WEIGHTS = False: its source tensors were not kept")
        return dict(self._trunk_sd)

    @staticmethod
    def _hooks(transformer_options):
        """-> the options dict if it carries Python hooks the executor has to call, else None (fast path).  The two hooks whose contract is a
        torch.nn.Module (block_inner_modifiers get `layer` and the block, group_norm_wrapper gets the GroupNorm) receive the stand-ins
        defined at the top of this file."""
        to = transformer_options
        if not to:
            return None
        if to.get("patches") or to.get("patches_replace") or to.get("block_modifiers") or to.get("block_inner_modifiers") or to.get("group_norm_wrapper") is not None:
            return to
        return None

    # reference-compatible signature (backend/nn/unet.py:696): NCHW in, NCHW out
    def forward(self, x, timesteps=None, context=None, y=None, control=None, transformer_options=None, **kwargs):
        assert (y is not None) == (self.num_classes is not None)
        bu, c, hh, ww = x.shape
        ctxc = self.prepare_context(context, y)
        ones = torch.zeros(bu, dtype=torch.float32, device=self.device)  # sigma = 0 -> scale 1/sqrt(0 + 1) = 1
        x = x.to(device=self.device, dtype=torch.float32)
        concat_term = None
        if self.concat_channels:  # the reference's forward gets latent and concat channels already concatenated
            concat_term = self.prepare_concat(x[:, self.latent_channels:].contiguous(), bu)
            x = x[:, :self.latent_channels]
        xcol = ops.unet_pack_input(x.contiguous(), ones, 1, 1.0)
        eps = self.forward_packed(xcol, timesteps.to(device=self.device, dtype=torch.float32).contiguous(), ctxc, bu, hh, ww, control,
                                  transformer_options, concat_term)
        return eps.view(bu, hh, ww, -1).permute(0, 3, 1, 2).to(x.dtype)

    __call__ = forward
