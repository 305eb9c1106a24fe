"""MI355X-native `ControlNet` -- drop-in for backend/nn/cnets/cldm.py:5-270 (same class name, same LDM / ControlNet checkpoint keys,
`forward(x, hint, timesteps, context, y)` -> list of residuals: one per input block (through its zero conv) + the middle block).

The trunk IS the UNet encoder (cldm.py:101-204 enumerates it with the UNet's own block classes), so it runs on the UNet executor
(../unet.py: same flat layout, same kernels, fp16 NHWC activations, cached cross-attention K / V^T and label embedding).  What is
added: `input_hint_block` (8 small 3x3 convs on the pixel-space hint, channels zero-padded to the GEMM's 64-channel granule) and the 1x1
`zero_convs` / `middle_block_out` (plain GEMMs on NHWC).

The guided hint depends on the hint image only -- not on x, t or the conditioning (cldm.py:232 feeds `emb` / `context` to a stack of
plain convs that ignore them) -- so it is computed ONCE per hint and cached, where the reference recomputes it every step.
Residuals are returned as NCHW *views* of the NHWC fp16 buffers (no transposes); the UNet executor adds such channels-last
residuals with a plain elementwise kernel."""
import torch

from .... import hipops as ops
from ..layout import HINT_BLOCK
from ..unet import IntegratedUNet2DConditionModel, _conv_w


def _pad64(c):
    return -(-c // 64) * 64


class ControlNet(IntegratedUNet2DConditionModel):
    encoder_only = True
    TRUNK_PREFIXES = ()  # nothing builds a Control-LoRA on top of a ControlNet: do not keep its source tensors alive

    def __init__(self, config, state_dict, device="cuda", hint_channels=3, arena_bytes=None):
        self.hint_channels = hint_channels
        if hint_channels * 9 > 64:
            raise NotImplementedError("hint_channels * 9 must be <= 64 (im2col'ed first conv)")
        super().__init__(config, state_dict, device=device, arena_bytes=arena_bytes)
        self._hint_key, self._guided_hint, self._hint_ref = None, None, None

    def _load_extra(self, sd, w):
        dev = self.device

        def T(key):
            return sd[key].to(device=dev, dtype=torch.float16).contiguous()
        # input_hint_block: conv i reads cin_pad channels and writes cout_pad (zero rows / columns for the padding; SiLU(0) = 0 keeps them 0)
        chain, cin = [], self.hint_channels
        for i, (cout, stride) in enumerate(HINT_BLOCK + ((self.layout.model_channels, 1),)):
            wt, bs = sd[f"input_hint_block.{2 * i}.weight"].to(dev, torch.float16), T(f"input_hint_block.{2 * i}.bias")
            cout_p = _pad64(cout)
            if i == 0:
                cw = _conv_w(wt)                                   # [cout, 9 * hint_channels] for the im2col'ed first conv
                wp = cw.new_zeros(cout_p, 64)
                wp[:cout, :cw.shape[1]] = cw
            else:
                cin_p = _pad64(cin)
                full = wt.new_zeros(cout_p, cin_p, 3, 3)
                full[:cout, :cin] = wt
                wp = _conv_w(full)
            bp = bs.new_zeros(cout_p)
            bp[:cout] = bs
            chain.append((wp.contiguous(), bp.contiguous(), stride, cout_p))
            cin = cout
        w["hint"] = chain
        w["zero_convs"] = [(T(f"zero_convs.{i}.0.weight").reshape(c, c).contiguous(), T(f"zero_convs.{i}.0.bias"))
                           for i, c in enumerate(self.layout.zero_conv_channels)]
        co = self.layout.out_ch
        w["middle_block_out"] = (T("middle_block_out.0.weight").reshape(co, co).contiguous(), T("middle_block_out.0.bias"))

    # ---- hint (once per hint image) -----------------------------------------------------------------------------------------------
    def guided_hint(self, hint):
        """hint [B, hint_channels, 8h, 8w] (any float dtype) -> fp16 NHWC [B, h, w, model_channels]; cached on tensor identity."""
        try:
            ver = hint._version
        except RuntimeError:   # inference-mode tensors carry no version counter
            ver = -1
        key = (hint.data_ptr(), tuple(hint.shape), hint.dtype, ver)
        if self._hint_key == key and self._hint_ref is hint:
            return self._guided_hint
        b, c, hh, ww = hint.shape
        x = hint.to(device=self.device, dtype=torch.float32).contiguous()
        chain = self.w["hint"]
        xl = ops.vae_pack_latent(x, 1.0, 0.0, ld=8)                 # NCHW fp32 -> NHWC fp16, zero padded to 8 channels
        wp, bp, _, cout = chain[0]
        h = ops.linear(ops.im2col3x3_smallc(xl, c), wp, bp).view(b, hh, ww, cout)
        for wp, bp, stride, cout in chain[1:]:
            h = ops.silu(h)
            bb, h2, w2, _ = h.shape
            oh, ow = (h2 + 2 - 3) // stride + 1, (w2 + 2 - 3) // stride + 1
            h = ops.conv_gemm(h, wp, cout, kh=3, stride=stride, pad=1, bias=bp).view(bb, oh, ow, cout)
        self._hint_key, self._guided_hint = key, h.clone()          # out of the arena: it lives across steps
        self._hint_ref = hint   # keeps the image alive: a later hint cannot land on its address (and its key) while this entry is cached
        self._hint_keepalive = hint
        return self._guided_hint

    # ---- per step ------------------------------------------------------------------------------------------------------------------
    def _forward_impl(self, xcol, t, ctxc, bu, hh, ww, arena, guided_hint):  # noqa: the ControlNet trunk has its own walk
        lay = self.layout
        t_emb = ops.timestep_embedding(t, lay.model_channels)
        e1 = ops.linear(t_emb, *self.w["te0"])
        e1 = ops.silu(e1, out=e1)
        emb = ops.linear(e1, *self.w["te2"], residual=ctxc.label)
        emb_all = ops.linear(ops.silu(emb), *self.w["emb_all"])
        assert guided_hint.shape[0] == bu
        outs = []
        h = None
        for bi, blk in enumerate(lay.input_blocks):
            if bi == 0:
                cw, cb = self.w[blk[0].key]
                # h = conv_in(x) + guided_hint (cldm.py:243-246): the hint rides in as the GEMM's residual operand
                h = ops.linear(xcol, cw, cb, residual=guided_hint.view(-1, lay.model_channels)).view(bu, hh, ww, lay.model_channels)
            else:
                h = self._run_block(blk, h, None, emb_all, ctxc, arena)
            zw, zb = self.w["zero_convs"][bi]
            outs.append(ops.linear(h.view(-1, h.shape[-1]), zw, zb).view(h.shape))
        h = self._run_block(lay.middle, h, None, emb_all, ctxc, arena)
        outs.append(ops.linear(h.view(-1, h.shape[-1]), *self.w["middle_block_out"]).view(h.shape))
        return outs

    def hint_for_batch(self, hint, bu):
        """The guided hint repeated to the network batch; cached with the guided hint itself (one tensor per hint image and batch)."""
        gh = self.guided_hint(hint)
        if gh.shape[0] == bu:
            return gh
        if bu % gh.shape[0] != 0:
            raise ValueError(f"hint batch {gh.shape[0]} does not divide the UNet batch {bu}")
        cached = getattr(self, "_gh_rep", None)
        if cached is None or cached[0] is not gh or cached[1] != bu:
            self._gh_rep = cached = (gh, bu, gh.repeat(bu // gh.shape[0], 1, 1, 1))
        return cached[2]

    def forward_static(self, xcol, t, ctxc, bu, hh, ww, gh):
        """The trunk on inputs that are already packed (`fmx_unet_pack_input` columns, fp32 timesteps, cached conditioning, guided hint of the
        network batch): kernels and arena allocations only, so `KModel` can capture it in its graph.  -> residuals as in `forward`."""
        from ....runtime import ArenaOverflow
        while True:
            arena = self._get_arena(bu, hh, ww)
            arena.reset()
            try:
                with arena:
                    outs = self._forward_impl(xcol, t, ctxc, bu, hh, ww, arena, gh)
                break
            except ArenaOverflow:
                torch.cuda.synchronize(self.device)
                self._arena_bytes = arena.capacity * 2
                self._arena = None
        return [o.permute(0, 3, 1, 2) for o in outs]

    def forward(self, x, hint, timesteps, context, y=None, **kwargs):
        """Reference signature (cldm.py:229): x [B, C, h, w] (already scaled by calculate_input), hint [B or 1, hc, 8h, 8w], timesteps [B],
        context [B, T, Dc], y [B, adm] | None -> list of [B, C_i, h_i, w_i] fp16 residuals (channels-last memory), valid until the next call."""
        assert (y is not None) == (self.num_classes is not None)
        bu, c, hh, ww = x.shape
        ctxc = self.prepare_context(context, y)
        gh = self.hint_for_batch(hint, bu)
        if gh.shape[1] != hh or gh.shape[2] != ww:
            raise ValueError(f"hint {tuple(hint.shape)} is not 8x the latent {tuple(x.shape)}")
        zero = torch.zeros(bu, dtype=torch.float32, device=self.device)  # sigma = 0 -> the pack kernel's 1/sqrt(sigma^2 + 1) is 1
        xcol = ops.unet_pack_input(x.to(device=self.device, dtype=torch.float32).contiguous(), zero, 1, 1.0)
        t = timesteps.to(device=self.device, dtype=torch.float32).contiguous()
        return self.forward_static(xcol, t, ctxc, bu, hh, ww, gh)

    __call__ = forward
