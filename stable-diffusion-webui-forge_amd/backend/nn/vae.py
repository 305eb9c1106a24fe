"""MI355X-native AutoencoderKL: the decoder (txt2img path) and the encoder (img2img / hires path).

Drop-in for `IntegratedAutoencoderKL.decode / process_out` (reference: backend/nn/vae.py:305-316, Decoder.forward
:248-271) plus the `VAE.decode_inner` post-processing (backend/patcher/vae.py:128-148): fp16 NHWC activations, all
3x3 convs on the MFMA implicit-GEMM kernel with the nearest-x2 Upsample (vae.py:35-57) fused into the conv's
loader, GroupNorm(eps 1e-6)+SiLU fused, residual adds in the conv epilogue.  The single-head mid-block attention
(C = 512, N up to 16 384; attention.py:412-422) runs as S = QK^T (GEMM) -> row softmax -> PV (GEMM): with 288 GB
of HBM the N x N score matrix of one image (512 MB at 1024^2) is affordable and keeps the MFMA GEMM as the only
heavy kernel; V's bias is added after PV (softmax rows sum to 1).
Element type (`dtype`): the reference decodes in bf16 wherever the part supports it and in fp32 otherwise, `--vae-in-fp16` / `--vae-in-bf16`
select by hand (backend/memory_management.py:190-205 VAE_DTYPES, :840-855 vae_dtype(); forge_amd.backend.memory_management mirrors the
chooser).  Here: fp16 (default: 8x finer than bf16, the type the parity fixtures and the bench run in) or bfloat16 -- the same kernels built for
both (libfmx ABI 6) -- with fp32 accumulation everywhere (GEMM, norm statistics, softmax).  Trained SDXL VAE weights leave fp16's range in the
decoder's upper levels; an fp16 decode whose output is not finite is therefore repeated in bfloat16 and the executor stays there
(`auto_bf16_fallback`, one counting kernel over the output per decode; tests/golden/tiny_vae_overflow.pt is such a decoder).

Encoder (`encode`, reference Encoder.forward vae.py:183-200, Downsample :60-74, quant_conv + DiagonalGaussianDistribution
:16-29, :296-303): same kernels; the Downsample's right/bottom-only zero padding costs nothing -- the conv loader's bounds
check already returns zeros for taps past the input, so only the output extent is passed (pad = 0, stride 2).  Present only
when the state dict carries `encoder.*` keys.
"""
import torch

from ... import hipops as ops
from ...runtime import Arena, ArenaOverflow
from .layout import vae_decoder_layout, vae_encoder_layout
from .unet import _conv_w


class DiagonalGaussianDistribution:
    """nn/vae.py:14-30: what a `model_vae_regulation` hook is handed (mean | logvar moments, NCHW fp32) -- same attributes and methods."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)   # CPU default generator, as the reference (:27)

    def mode(self):
        return self.mean


class IntegratedAutoencoderKL:
    def __init__(self, config, state_dict, device="cuda", dtype=torch.float16, auto_bf16_fallback=True):
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError(f"VAE element type {dtype}: the native decoder is built for float16 and bfloat16 (the reference's fp32 VAE has no "
                                      f"MFMA form on gfx950; bfloat16 has fp32's range)")
        self.dtype = dtype
        self.auto_bf16_fallback = bool(auto_bf16_fallback)
        self.fallbacks = 0          # decodes repeated in bfloat16 because the fp16 result was not finite
        self.up2x_trace = set()     # Upsample layers that ran as four phase convolutions in the last decode (told to the rounding oracle, tests only)
        self.tap = None             # test hook: a dict that receives every layer's stored output of the next decode (see _tap)
        # by reference, and only while a fallback can still happen (an fp16 executor with the guard on): the bfloat16 weights are made from it then, and
        # it is released -- together with the fp16 copy, which no later decode uses -- as soon as they exist (the caller's full state dict,
        # often fp32 and already on the device, used to stay alive for the executor's lifetime)
        self._source = state_dict if (dtype == torch.float16 and self.auto_bf16_fallback) else None
        self._weights = {}
        self.config = dict(config)
        self.layout = vae_decoder_layout(config)
        self.device = torch.device(device)
        self.scaling_factor = self.layout.scaling_factor
        self.shift_factor = self.layout.shift_factor
        self.latent_channels = self.layout.latent_channels
        self._arena = None
        self.up_factor = 2 ** (len(self.layout.levels) - 1)
        self.has_encoder = "encoder.conv_in.weight" in state_dict
        self.enc_layout = vae_encoder_layout(config) if self.has_encoder else None
        self.w = self._weights[dtype] = self._load(state_dict, dtype)

    def set_dtype(self, dtype):
        """Switch the element type of every later decode / encode (weights of the new type are converted from the source state dict once)."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError(f"VAE element type {dtype}")
        if dtype not in self._weights:
            if self._source is not None:
                self._weights[dtype] = self._load(self._source, dtype)
            else:
                # the source was released: re-round the resident copy.  fp16 -> bf16 keeps every exponent and drops 3 mantissa bits (what a load
                # from an fp16 checkpoint gives).  bf16 -> fp16 would start from 8-bit mantissas AND re-round the folded attention bias instead of
                # recomputing it in fp32 -- silently less accurate than `_load(sd, fp16)` -- so it is refused: build the VAE in fp16 (with the bf16
                # guard, which keeps the source while it may be needed) or hand the state dict again.
                src_dt, src = next(iter(self._weights.items()))
                if src_dt == torch.bfloat16 and dtype == torch.float16:
                    raise RuntimeError("set_dtype(float16) on a VAE whose only resident weights are bfloat16: the fp32 source was released; "
                                       "construct IntegratedAutoencoderKL(..., dtype=torch.float16) from the state dict instead")
                self._weights[dtype] = {k: (tuple(t.to(dtype) for t in v) if isinstance(v, tuple) else v.to(dtype) if torch.is_tensor(v) else v) for k, v in src.items()}
                for k in [k for k in self._weights[dtype] if k.endswith(".up2x")]:   # tap sums: from the re-rounded taps, one rounding (not a second one of the fp16 sums)
                    base = self._weights[dtype][k[:-len(".up2x")]][0]
                    self._weights[dtype][k] = ops.fold_up2x_weights(base, base.shape[1] // 9)
        self.dtype, self.w = dtype, self._weights[dtype]

    def _load(self, sd, dt):
        dev = self.device

        def T(k):
            return sd[k].to(device=dev, dtype=dt).contiguous()

        def conv(k):
            return (_conv_w(sd[k + ".weight"].to(dev, dt)), T(k + ".bias"))

        def norm(k):
            return (T(k + ".weight"), T(k + ".bias"))

        w = {}
        lay = self.layout
        lc = lay.latent_channels
        # post_quant_conv (1x1, lc->lc) folded into conv_in's im2col GEMM is not exact at the borders (zero padding
        # happens after the 1x1 conv + bias), so it runs as its own tiny GEMM on a 64-wide zero-padded latent.
        if lay.use_post_quant_conv:
            pw = sd["post_quant_conv.weight"].to(dev, dt).reshape(lc, lc)
            wp = pw.new_zeros(8, 64)
            wp[:lc, :lc] = pw
            bp = pw.new_zeros(8)
            bp[:lc] = T("post_quant_conv.bias")
            w["pq"] = (wp.contiguous(), bp.contiguous())
        full = sd["decoder.conv_in.weight"].to(dev, dt)               # [block_in, lc, 3, 3]
        if lc * 9 <= 64:
            cw = _conv_w(full)
            wp = cw.new_zeros(cw.shape[0], 64)
            wp[:, :cw.shape[1]] = cw
        else:
            # 16-channel latents (Flux / SD3 VAE): the 3x3 conv runs as the regular implicit GEMM on the 64-wide zero-padded latent
            if lay.use_post_quant_conv:
                raise NotImplementedError("post_quant_conv with more than 7 latent channels")
            padded = full.new_zeros(full.shape[0], 64, 3, 3)
            padded[:, :lc] = full
            wp = _conv_w(padded)
        w["conv_in"] = (wp.contiguous(), T("decoder.conv_in.bias"))

        def res(k, cin, cout):
            w[k + ".n1"], w[k + ".c1"] = norm(k + ".norm1"), conv(k + ".conv1")
            w[k + ".n2"], w[k + ".c2"] = norm(k + ".norm2"), conv(k + ".conv2")
            if cin != cout:
                w[k + ".sc"] = conv(k + ".nin_shortcut")

        bi = lay.block_in
        res("decoder.mid.block_1", bi, bi)
        res("decoder.mid.block_2", bi, bi)
        a = "decoder.mid.attn_1"
        w[a + ".norm"] = norm(a + ".norm")
        qw, qb = conv(a + ".q")
        kw, kb = conv(a + ".k")
        w[a + ".qk"] = (torch.cat([qw, kw], 0).contiguous(), torch.cat([qb, kb], 0).contiguous())
        w[a + ".v"] = conv(a + ".v")
        w[a + ".proj_out"] = conv(a + ".proj_out")
        w[a + ".proj_out_vbias"] = self._fold_v_bias(w[a + ".proj_out"], w[a + ".v"][1])
        for _, blocks, up in lay.levels:
            for key, cin, cout in blocks:
                res(key, cin, cout)
            if up is not None:
                w[up] = conv(up + ".conv")
                w[up + ".up2x"] = ops.fold_up2x_weights(w[up][0], w[up][0].shape[1] // 9)   # the Upsample convolution as four phase convolutions (hipops.conv3x3_up2x)
        w["norm_out"] = norm("decoder.norm_out")
        w["conv_out"] = conv("decoder.conv_out")
        if self.has_encoder:
            el = self.enc_layout
            cw = _conv_w(sd["encoder.conv_in.weight"].to(dev, dt))   # [ch, 9*in_channels]
            if cw.shape[1] > 64:
                raise NotImplementedError("encoder in_channels*9 must be <= 64 (im2col'ed first conv)")
            wp = cw.new_zeros(cw.shape[0], 64)
            wp[:, :cw.shape[1]] = cw
            w["e.conv_in"] = (wp.contiguous(), T("encoder.conv_in.bias"))
            for _, blocks, down in el.levels:
                for key, cin, cout in blocks:
                    res(key, cin, cout)
                if down is not None:
                    w[down] = conv(down + ".conv")
            ebi = el.block_in
            res("encoder.mid.block_1", ebi, ebi)
            res("encoder.mid.block_2", ebi, ebi)
            a = "encoder.mid.attn_1"
            w[a + ".norm"] = norm(a + ".norm")
            qw, qb = conv(a + ".q")
            kw, kb = conv(a + ".k")
            w[a + ".qk"] = (torch.cat([qw, kw], 0).contiguous(), torch.cat([qb, kb], 0).contiguous())
            w[a + ".v"] = conv(a + ".v")
            w[a + ".proj_out"] = conv(a + ".proj_out")
            w[a + ".proj_out_vbias"] = self._fold_v_bias(w[a + ".proj_out"], w[a + ".v"][1])
            w["e.norm_out"] = norm("encoder.norm_out")
            w["e.conv_out"] = conv("encoder.conv_out")
            if el.use_quant_conv:
                qc = sd["quant_conv.weight"].to(dev, dt).reshape(2 * lc, 2 * lc)
                qp = qc.new_zeros(2 * lc, 64)  # the moments travel in a 64-wide zero-padded NHWC buffer (GEMM K tile)
                qp[:, :2 * lc] = qc
                w["e.quant"] = (qp.contiguous(), T("quant_conv.bias"))
        torch.cuda.synchronize(dev)
        return w

    @staticmethod
    def _fold_v_bias(proj_out, b_v):
        """proj_out(P V + b_v) = W_o (P V) + (W_o b_v + b_o): the value bias of the mid-block attention (vae.py:118-137; softmax rows sum to 1, so it
        is the same vector for every token) folded into proj_out's bias once at load time, in fp32."""
        w_o, b_o = proj_out
        return (b_o.float() + w_o.float() @ b_v.float()).to(b_o.dtype).contiguous()

    # ------------------------------------------------------------------------------------------------------------
    def _tap(self, key, t, nchw=True):
        """test hook (tests/test_gpu_vae_sharp_parity.py): with `self.tap` a dict, every layer's stored output is copied out as fp32 NCHW"""
        if self.tap is not None:
            self.tap[key] = (t.permute(0, 3, 1, 2) if nchw else t).float().cpu()

    def _res(self, k, x, cin, cout, arena):
        b, hh, ww, _ = x.shape
        out = ops.empty((b, hh, ww, cout), self.dtype)
        out_part = ops.stats_buffer(b, hh * ww, cout)
        m = arena.mark()
        # the decoder's full-resolution level (128 output channels): norm -> swish -> conv as ONE launch each, the normalised tensor never stored
        # (csrc/fmx_conv_patch.hip; same values, same rounding sites as the two launches below)
        fused = ops.conv3x3_gn_silu_eligible(x, cout)
        if fused:
            h, h_st = ops.conv3x3_gn_silu(x, *self.w[k + ".n1"], 1e-6, *self.w[k + ".c1"])
        else:
            g1 = ops.groupnorm(x, *self.w[k + ".n1"], 1e-6, silu=True)          # statistics: left on x by the GEMM that produced it
            h, h_st = ops.conv_gemm(g1, self.w[k + ".c1"][0], cout, kh=3, pad=1, bias=self.w[k + ".c1"][1], stats=True)
        self._tap(k + ".h", h.view(b, hh, ww, cout))
        sk = ops.conv_gemm(x, self.w[k + ".sc"][0], cout, bias=self.w[k + ".sc"][1]) if cin != cout else x.view(-1, cout)
        if fused and h_st is not None:
            _, st = ops.conv3x3_gn_silu(h.view(b, hh, ww, cout), *self.w[k + ".n2"], 1e-6, *self.w[k + ".c2"], residual=sk, out=out.view(-1, cout),
                                        stats=h_st, stats_partial=out_part)
        else:
            g2 = ops.groupnorm(h.view(b, hh, ww, cout), *self.w[k + ".n2"], 1e-6, silu=True, stats=h_st)
            _, st = ops.conv_gemm(g2, self.w[k + ".c2"][0], cout, kh=3, pad=1, bias=self.w[k + ".c2"][1], residual=sk, out=out.view(-1, cout), ld_out=cout,
                                  stats=True, stats_partial=out_part)
        arena.release(m)
        self._tap(k, out)
        return ops.attach_stats(out, st)

    def _attn(self, x, arena, a="decoder.mid.attn_1"):
        b, hh, ww, c = x.shape
        n = hh * ww
        out = ops.empty((b, hh, ww, c), self.dtype)
        out_part = ops.stats_buffer(b, n, c)
        m = arena.mark()
        g = ops.groupnorm(x, *self.w[a + ".norm"], 1e-6).view(-1, c)
        qk = ops.linear(g, *self.w[a + ".qk"])                                  # [B*N, 2C]
        o = ops.empty((b * n, c), self.dtype)
        npad = -(-n // 64) * 64
        scale = c ** -0.5
        if c == 512:
            # fused: one launch for the whole batch, scores never leave the chip (csrc/fmx_attention512.hip).  V^T for all images comes from
            # ONE operand-swapped GEMM when the token count needs no padding; V's bias rides in proj_out's (softmax rows sum to 1)
            vt = ops.empty((c, b * npad), self.dtype)
            if npad == n:
                ops.conv_gemm(self.w[a + ".v"][0], g, b * n, out=vt, ld_out=b * n)
            else:
                vt.zero_()  # padded key columns must hold finite values (their probabilities are exactly zero)
                for bi in range(b):
                    ops.conv_gemm(self.w[a + ".v"][0], g[bi * n:(bi + 1) * n], n, out=vt[:, bi * npad:], ld_out=b * npad)
            ops.attention_single_head512(qk, qk[:, c:], vt, o, batch=b, nq=n, nk=n, nk_pad=npad, q_bs=n * 2 * c, q_rs=2 * c, k_bs=n * 2 * c, k_rs=2 * c,
                                         vt_bs=npad, vt_ds=b * npad, scale=scale)
            if self.tap is not None:
                self._tap(a + ".q", qk[:, :c].reshape(b, hh, ww, c))
                self._tap(a + ".k", qk[:, c:].reshape(b, hh, ww, c))
                self._tap(a + ".v", vt.view(c, b, npad)[:, :, :n].permute(1, 0, 2), nchw=False)    # V^T without its bias, [B, C, N]
                self._tap(a + ".o", o.view(b, hh, ww, c))
            _, st = ops.linear(o, self.w[a + ".proj_out"][0], self.w[a + ".proj_out_vbias"], residual=x.view(-1, c), out=out.view(-1, c), ld_out=c,
                               n=b, h=hh, w=ww, stats=True, stats_partial=out_part)
            arena.release(m)
            self._tap(a, out)
            return ops.attach_stats(out, st)
        # other widths: S = scale * Q K^T (GEMM) -> row softmax -> P V (GEMM) per image
        for bi in range(b):
            mk = arena.mark()
            gb = g[bi * n:(bi + 1) * n]
            q = qk[bi * n:(bi + 1) * n, :c]
            kk = qk[bi * n:(bi + 1) * n, c:]
            s = ops.empty((n, npad), self.dtype)
            if npad != n:
                s.zero_()  # padded key columns must hold finite values for the PV GEMM (their P is never produced)
            ops.conv_gemm(q, kk, n, alpha=scale, out=s, ld_out=npad)             # S = scale * Q K^T
            ops.softmax_rows_(s[:, :n])
            if npad != n:
                s[:, n:].zero_()
            vt = ops.empty((c, npad), self.dtype)
            if npad != n:
                vt.zero_()
            ops.conv_gemm(self.w[a + ".v"][0], gb, n, out=vt, ld_out=npad)        # V^T (bias deferred)
            ops.conv_gemm(s, vt, c, bias=self.w[a + ".v"][1], out=o[bi * n:(bi + 1) * n], ld_out=c)  # P V + b_v
            arena.release(mk)
        if self.tap is not None:
            self._tap(a + ".q", qk[:, :c].reshape(b, hh, ww, c))
            self._tap(a + ".k", qk[:, c:].reshape(b, hh, ww, c))
            self._tap(a + ".o", o.view(b, hh, ww, c))
        _, st = ops.linear(o, *self.w[a + ".proj_out"], residual=x.view(-1, c), out=out.view(-1, c), ld_out=c, n=b, h=hh, w=ww, stats=True,
                           stats_partial=out_part)
        arena.release(m)
        self._tap(a, out)
        return ops.attach_stats(out, st)

    def _decode_impl(self, z, arena):
        """z fp32 NCHW [B, lc, h, w] (already process_out'ed) -> fp16 [B*8h*8w, 4] (first out_channels valid)"""
        lay = self.layout
        b, lc, hh, ww = z.shape
        zl = ops.vae_pack_latent(z, 1.0, 0.0, ld=64, dtype=self.dtype)                              # [B,h,w,64] zero padded
        if lay.use_post_quant_conv:
            zq = ops.conv_gemm(zl.view(-1, 64), self.w["pq"][0], 8, bias=self.w["pq"][1]).view(b, hh, ww, 8)
        else:
            zq = zl
        if lc * 9 <= 64:
            h, st = ops.linear(ops.im2col3x3_smallc(zq, lc), *self.w["conv_in"], n=b, h=hh, w=ww, stats=True)
        else:
            h, st = ops.conv_gemm(zq, self.w["conv_in"][0], lay.block_in, kh=3, pad=1, bias=self.w["conv_in"][1], stats=True)
        h = ops.attach_stats(h.view(b, hh, ww, lay.block_in), st)
        self._tap("conv_in", h)
        bi = lay.block_in
        h = self._res("decoder.mid.block_1", h, bi, bi, arena)
        h = self._attn(h, arena)
        h = self._res("decoder.mid.block_2", h, bi, bi, arena)
        for _, blocks, up in lay.levels:
            for key, cin, cout in blocks:
                h = self._res(key, h, cin, cout, arena)
            if up is not None:
                bb, h2, w2, c = h.shape
                if ops.conv3x3_up2x_eligible(h, c):
                    h, st = ops.conv3x3_up2x(h, self.w[up + ".up2x"], self.w[up][1], c)
                    self.up2x_trace.add(up)
                else:
                    h, st = ops.conv_gemm(h, self.w[up][0], c, kh=3, pad=1, up=(2 * h2, 2 * w2), bias=self.w[up][1], stats=True)
                h = ops.attach_stats(h.view(bb, 2 * h2, 2 * w2, c), st)
                self._tap(up, h)
        co, cin = lay.out_channels, h.shape[-1]
        nb, oh, ow = h.shape[0], h.shape[1], h.shape[2]
        if co <= 4 and cin % 32 == 0 and ops._CONV_GN_FUSE and ops._attached_stats(h) is not None and nb * oh * ow >= (1 << 16):
            # norm_out -> swish -> conv_out in ONE launch (round 6): the normalisation applied while the direct kernel stages its input patch, the
            # GroupNorm apply pass (one read + one write of the full-resolution tensor) gone; same values, same rounding sites
            y = ops.conv3x3_narrow_gn_silu(h, *self.w["norm_out"], 1e-6, self.w["conv_out"][0], self.w["conv_out"][1], co)
        else:
            g = ops.groupnorm(h, *self.w["norm_out"], 1e-6, silu=True)
            if co <= 4 and cin % 32 == 0 and g.numel() * 2 < 3.0e9:
                # conv_out as a DIRECT 3x3 kernel (round 4): the implicit GEMM spends 2.2 ms per 8 x 1024^2 on a 3-column output (nine-fold im2col gather,
                # 97 % padding columns); this one stages each input patch once and writes all four columns of [npix, 4] (the pad column as zeros)
                y = ops.conv3x3_narrow(g, self.w["conv_out"][0], self.w["conv_out"][1], co)
            else:
                # [npix, 4] with 3 valid columns: the GEMM epilogue never writes column 3, and the arena hands out recycled bytes -- zeroed, so that the
                # overflow guard's scan of the whole buffer (ops.count_nonfinite) cannot trip over a stale inf / NaN half-word there
                y = ops.empty((nb * oh * ow, 4), self.dtype)
                y.zero_()
                y = ops.conv_gemm(g, self.w["conv_out"][0], co, kh=3, pad=1, bias=self.w["conv_out"][1], out=y, ld_out=4)
        if self.tap is not None:
            self._tap("conv_out", y.view(nb, oh, ow, 4)[..., :co])
        return y

    def _run(self, z):
        b, lc, hh, ww = z.shape
        f = self.up_factor
        need = max(1 << 28, int(b * hh * ww * f * f * self.layout.final_ch * 2 * 14) + 2 * (hh * ww) ** 2 * 2)
        while True:
            if self._arena is None or self._arena.capacity < need:
                self._arena = None
                self._arena = Arena(need, self.device)
            arena = self._arena
            arena.reset()
            try:
                with arena:
                    y = self._decode_impl(z, arena)
                if self.dtype == torch.float16 and self.auto_bf16_fallback and ops.count_nonfinite(y) > 0:
                    self._fall_back_to_bf16("decode")
                    continue
                return y
            except ArenaOverflow:
                torch.cuda.synchronize(self.device)
                need = arena.capacity * 2
                self._arena = None

    def _fall_back_to_bf16(self, what):
        import warnings
        self.fallbacks += 1
        warnings.warn(f"VAE {what}: the float16 result is not finite (activations beyond 65504, as with trained SDXL VAE weights); repeating in "
                      f"bfloat16 and keeping this VAE in bfloat16 from here on (construct it with dtype=torch.bfloat16 to start there)")
        self.set_dtype(torch.bfloat16)
        # this VAE stays in bfloat16: neither the fp16 weights nor the caller's state dict are needed again
        self._weights.pop(torch.float16, None)
        self._source = None

    # ---- encoder -----------------------------------------------------------------------------------------------------
    def _encode_impl(self, x, arena):
        """x fp32 NCHW [B, 3, H, W] in [-1, 1] -> moments fp16 [B*h*w, 64] (first 2*lc columns valid: mean | logvar)"""
        el = self.enc_layout
        b, c, hh, ww = x.shape
        xl = ops.vae_pack_latent(x, 1.0, 0.0, ld=8, dtype=self.dtype)                               # NHWC fp16, zero padded to 8 channels
        col = ops.im2col3x3_smallc(xl, c)
        h, st = ops.linear(col, *self.w["e.conv_in"], n=b, h=hh, w=ww, stats=True)
        h = ops.attach_stats(h.view(b, hh, ww, el.ch), st)
        for _, blocks, down in el.levels:
            for key, cin, cout in blocks:
                h = self._res(key, h, cin, cout, arena)
            if down is not None:
                bb, h2, w2, cc = h.shape
                oh, ow = (h2 + 1 - 3) // 2 + 1, (w2 + 1 - 3) // 2 + 1            # F.pad (0,1,0,1) then 3x3 stride 2 (vae.py:67-70)
                h, st = ops.conv_gemm(h, self.w[down][0], cc, kh=3, stride=2, pad=0, out_hw=(oh, ow), bias=self.w[down][1], stats=True)
                h = ops.attach_stats(h.view(bb, oh, ow, cc), st)
        ebi = el.block_in
        h = self._res("encoder.mid.block_1", h, ebi, ebi, arena)
        h = self._attn(h, arena, "encoder.mid.attn_1")
        h = self._res("encoder.mid.block_2", h, ebi, ebi, arena)
        g = ops.groupnorm(h, *self.w["e.norm_out"], 1e-6, silu=True)
        npix = g.shape[0] * g.shape[1] * g.shape[2]
        lc2 = 2 * el.latent_channels
        mo = ops.empty((npix, 64), self.dtype)
        mo.zero_()
        ops.conv_gemm(g, self.w["e.conv_out"][0], lc2, kh=3, pad=1, bias=self.w["e.conv_out"][1], out=mo, ld_out=64)
        if el.use_quant_conv:
            mq = ops.empty((npix, 64), self.dtype)
            mq.zero_()
            ops.conv_gemm(mo, self.w["e.quant"][0], lc2, bias=self.w["e.quant"][1], out=mq, ld_out=64)
            mo = mq
        return mo

    def _run_encode(self, x):
        b, c, hh, ww = x.shape
        need = max(1 << 28, int(b * hh * ww * self.enc_layout.ch * 2 * 14) + 2 * ((hh // self.up_factor) * (ww // self.up_factor)) ** 2 * 2)
        while True:
            if self._arena is None or self._arena.capacity < need:
                self._arena = None
                self._arena = Arena(need, self.device)
            arena = self._arena
            arena.reset()
            try:
                with arena:
                    mo = self._encode_impl(x, arena)
                if self.dtype == torch.float16 and self.auto_bf16_fallback and ops.count_nonfinite(mo) > 0:
                    self._fall_back_to_bf16("encode")
                    continue
                return mo
            except ArenaOverflow:
                torch.cuda.synchronize(self.device)
                need = arena.capacity * 2
                self._arena = None

    def encode_moments(self, x):
        """-> fp32 NCHW [B, 2*lc, h, w] (mean | logvar): quant_conv(encoder(x)), vae.py:296-298."""
        if not self.has_encoder:
            raise RuntimeError("this AutoencoderKL was built without encoder weights")
        xf = x.to(device=self.device, dtype=torch.float32).contiguous()
        mo = self._run_encode(xf)
        b, _, hh, ww = x.shape
        f = self.up_factor
        lc2 = 2 * self.latent_channels
        return mo.view(b, hh // f, ww // f, 64)[..., :lc2].permute(0, 3, 1, 2).float()

    def encode(self, x, regulation=None, noise=None):
        """vae.py:296-303: posterior sample = mean + std * noise.  The reference draws torch.randn(shape) on the CPU default
        generator (vae.py:28); same here unless `noise` is given, so a seeded torch.manual_seed reproduces it."""
        if regulation is not None:
            # patcher/vae.py:166-178 `model_vae_regulation` (set by UnetPatcher.set_model_vae_regulation, patcher/base.py:155): the hook receives the
            # posterior and returns the latent.  The encoder runs natively; the posterior is the reference's value object over its moments.
            return regulation(DiagonalGaussianDistribution(self.encode_moments(x))).to(x.dtype)
        if not self.has_encoder:
            raise RuntimeError("this AutoencoderKL was built without encoder weights")
        xf = x.to(device=self.device, dtype=torch.float32).contiguous()
        b, _, hh, ww = x.shape
        f = self.up_factor
        lc = self.latent_channels
        if noise is None:
            noise = torch.randn(b, lc, hh // f, ww // f)
        noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
        mo = self._run_encode(xf)
        return ops.vae_sample_posterior(mo, 64, noise, lc).to(x.dtype)

    # ---- reference surface ---------------------------------------------------------------------------------------
    def process_in(self, latent):
        return (latent - self.shift_factor) * self.scaling_factor  # vae.py:312-313

    def process_out(self, latent):
        return (latent / self.scaling_factor) + self.shift_factor  # vae.py:315

    def _image_groups(self, b, hh, ww):
        """Images are independent: a batch whose decode arena (14 full-resolution fp16 tensors of `final_ch` channels per image) would not fit beside
        what is already resident is decoded in equal groups of whole images, each through the same arena (round 4: 64 x 1024^2 -- BASELINE config 4's
        global batch on ONE device -- asked for a 225 GiB arena).  The reference decodes one image at a time (modules/processing.py decode_latent_batch)."""
        f = self.up_factor
        per = hh * ww * f * f * self.layout.final_ch * 2 * 14
        try:
            free, total = torch.cuda.mem_get_info(self.device)
            budget = max(per, min(int(0.45 * total), int(0.8 * (free + (self._arena.capacity if self._arena is not None else 0)))))
        except Exception:  # noqa: BLE001
            budget = 64 << 30
        g = max(1, min(b, budget // max(per, 1)))
        while b % g:
            g -= 1
        return g

    def decode(self, z):
        """vae.py:305-311: z [B, lc, h, w] -> [B, 3, 8h, 8w] (same dtype as z)."""
        zf = z.to(device=self.device, dtype=torch.float32).contiguous()
        b, _, hh, ww = z.shape
        oc = self.layout.out_channels
        f = self.up_factor
        g = self._image_groups(b, hh, ww)
        if g == b:
            y = self._run(zf)
            return y.view(b, f * hh, f * ww, 4)[..., :oc].permute(0, 3, 1, 2).to(z.dtype)
        # every group runs through the SAME arena: its result must be copied out before the next group overwrites it (`.to(z.dtype)` makes no copy
        # when z already has the VAE's element type -- ADVICE r4: fp16 latents on an fp16 VAE came back as B/g views of the last group's arena)
        out = torch.empty(b, oc, f * hh, f * ww, dtype=z.dtype, device=self.device)
        for i in range(0, b, g):
            out[i:i + g].copy_(self._run(zf[i:i + g]).view(g, f * hh, f * ww, 4)[..., :oc].permute(0, 3, 1, 2))
        return out

    def decode_inner(self, samples_in):
        """patcher/vae.py:128-148: -> fp32 [B, 8h, 8w, 3] in [0, 1] (clamp((y+1)/2) fused into the unpack kernel)."""
        zf = samples_in.to(device=self.device, dtype=torch.float32).contiguous()
        b, _, hh, ww = samples_in.shape
        oc = self.layout.out_channels
        f = self.up_factor
        out = torch.empty(b, f * hh, f * ww, oc, dtype=torch.float32, device=self.device)
        g = self._image_groups(b, hh, ww)
        for i in range(0, b, g):          # (one group = the whole batch unless its arena would not fit: see _image_groups)
            y = self._run(zf[i:i + g])
            ops.vae_unpack_image(y, 4, g * f * f * hh * ww, oc, out[i:i + g])
        return out
