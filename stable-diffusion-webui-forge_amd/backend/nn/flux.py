"""MI355X-native executor for Forge's Flux transformer (MMDiT: 19 double-stream + 38 single-stream blocks in Flux.1-dev).

Drop-in for `IntegratedFluxTransformer2DModel` (reference: backend/nn/flux.py:310-418): same class name, same
`forward(x, timestep, context, y, guidance)` contract and the same checkpoint keys, but a flat walk over fused gfx950
kernels (include/fmx.h) instead of an nn.Module graph:

  * every Linear is the MFMA GEMM; bias, GELU-tanh (flux.py:193,280), the adaLN gate and the residual add
    (`x + gate * f(x)`, :254-262,301) are its epilogue, the `cat(attn, gelu(mlp))` of the single blocks (:299) is its
    two-source A operand -- no concat, no separate activation / gate / add kernels;
  * LayerNorm(no affine) + `(1 + scale) * . + shift` (:209-210,255,286,326) is one kernel;
  * per-head RMSNorm of q,k + rotary embedding + the `[B,L,3,H,D] -> [B,H,L,D]` permute (:43-49,115-139,217-247) is one
    HBM pass that writes q,k straight into the joint txt||img attention layout and V transposed for the fused attention
    kernel (flash-style, d_head 128), so `torch.cat((txt_q, img_q))` (:241-246) never happens;
  * all 19*2 + 38 + 1 adaLN modulation Linears depend only on `vec`: ONE GEMM per forward (the reference runs 77).
The rotary table depends only on the latent size and text length: built once on the host in float64 exactly as
`rope()` (:21-40) and cached.
fp16 or bf16 storage / MFMA operands with fp32 accumulation (`dtype`; the reference computes Flux in bf16): the bf16 entry points are the
same kernels compiled a second time with bfloat16 elements (csrc/fmx_common.hpp).
"""
import torch

from ... import hipops as ops


def _rope_table(ids, axes_dim, theta):
    """ids [L, 3] (float) -> fp32 [L, sum(axes)/2, 2] = (cos, sin); float64 frequencies as flux.py:21-40."""
    outs = []
    for i, dim in enumerate(axes_dim):
        scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        omega = 1.0 / (theta ** scale)
        ang = ids[:, i].double().unsqueeze(-1) * omega.unsqueeze(0)
        outs.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
    return torch.cat(outs, dim=1).float().contiguous()


class IntegratedFluxTransformer2DModel:
    def __init__(self, config, state_dict, device="cuda", dtype=torch.float16):
        """`dtype`: torch.float16 or torch.bfloat16 -- storage and MFMA operand type of the whole forward (accumulation, softmax, norms and
        epilogue math are fp32 either way).  bf16 is what the reference runs Flux.1 in; trained Flux weights overflow fp16 in the late
        single-stream blocks, so real checkpoints want bf16, while fp16 has 3 more mantissa bits for well-scaled activations."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("the Flux executor computes in fp16 or bf16")
        self.config = dict(config)
        self.device = torch.device(device)
        self.dtype = self.storage_dtype = self.computation_dtype = dtype
        self.hidden = config["hidden_size"]
        self.heads = config["num_heads"]
        self.head_dim = self.hidden // self.heads
        if self.head_dim != 128:
            raise NotImplementedError("the rotary / attention kernels are built for Flux's head_dim 128")
        self.mlp = int(self.hidden * config["mlp_ratio"])
        self.in_channels = config["in_channels"] * 4
        self.out_channels = self.in_channels
        self.guidance_embed = config["guidance_embed"]
        self.depth, self.depth_single = config["depth"], config["depth_single_blocks"]
        self.axes_dim, self.theta = list(config["axes_dim"]), config["theta"]
        self._pe_cache = {}
        self.tap = None    # test hook (tests/test_gpu_flux_sharp_parity.py): a dict that receives every stored stage output of the next forward
        self._load(state_dict)

    # ------------------------------------------------------------------------------------------------------------
    def _load(self, sd):
        dev = self.device

        def T(k):
            return sd[k].to(device=dev, dtype=self.dtype).contiguous()

        def lin(k):
            return T(k + ".weight"), (T(k + ".bias") if (k + ".bias") in sd else None)

        w = {}
        for k in ("img_in", "time_in.in_layer", "time_in.out_layer", "vector_in.in_layer", "vector_in.out_layer", "txt_in",
                  "final_layer.linear"):
            w[k] = lin(k)
        if self.guidance_embed:
            w["guidance_in.in_layer"], w["guidance_in.out_layer"] = lin("guidance_in.in_layer"), lin("guidance_in.out_layer")
        hs = self.hidden
        mod_w, mod_b, self._mod_off = [], [], {}
        off = 0

        def add_mod(key, k):
            nonlocal off
            ww, bb = lin(k)
            mod_w.append(ww)
            mod_b.append(bb)
            self._mod_off[key] = off
            off += ww.shape[0]

        for i in range(self.depth):
            b = f"double_blocks.{i}"
            for st in ("img", "txt"):
                add_mod(f"{b}.{st}", f"{b}.{st}_mod.lin")
                w[f"{b}.{st}.qkv"] = lin(f"{b}.{st}_attn.qkv")
                w[f"{b}.{st}.qs"] = T(f"{b}.{st}_attn.norm.query_norm.scale")
                w[f"{b}.{st}.ks"] = T(f"{b}.{st}_attn.norm.key_norm.scale")
                w[f"{b}.{st}.proj"] = lin(f"{b}.{st}_attn.proj")
                w[f"{b}.{st}.mlp0"] = lin(f"{b}.{st}_mlp.0")
                w[f"{b}.{st}.mlp2"] = lin(f"{b}.{st}_mlp.2")
        for i in range(self.depth_single):
            b = f"single_blocks.{i}"
            add_mod(b, f"{b}.modulation.lin")
            w1, b1 = lin(f"{b}.linear1")
            w[b + ".qkv"] = (w1[:3 * hs], b1[:3 * hs])          # row slices of one tensor: contiguous views
            w[b + ".mlp"] = (w1[3 * hs:], b1[3 * hs:])
            w[b + ".lin2"] = lin(f"{b}.linear2")
            w[b + ".qs"] = T(f"{b}.norm.query_norm.scale")
            w[b + ".ks"] = T(f"{b}.norm.key_norm.scale")
        add_mod("final", "final_layer.adaLN_modulation.1")
        w["mods"] = (torch.cat(mod_w, 0).contiguous(), torch.cat(mod_b, 0).contiguous())
        self._mod_total = off
        self.w = w
        torch.cuda.synchronize(dev)

    def _pe(self, h_len, w_len, ltxt):
        key = (h_len, w_len, ltxt)
        pe = self._pe_cache.get(key)
        if pe is None:
            ids = torch.zeros(ltxt + h_len * w_len, 3)
            ii = torch.zeros(h_len, w_len, 3)
            ii[..., 1] += torch.linspace(0, h_len - 1, steps=h_len)[:, None]   # flux.py:409-412
            ii[..., 2] += torch.linspace(0, w_len - 1, steps=w_len)[None, :]
            ids[ltxt:] = ii.reshape(-1, 3)
            pe = _rope_table(ids, self.axes_dim, self.theta).to(self.device)
            self._pe_cache = {key: pe}
        return pe

    # ------------------------------------------------------------------------------------------------------------
    def _tap(self, key, t):
        if self.tap is not None:
            self.tap[key] = t.float().cpu()

    def _tap_qkv(self, key, qj, kj, vtj, bsz, ltot, lpad):
        if self.tap is not None:
            self._tap(key + ".q", qj[:, :ltot])
            self._tap(key + ".k", kj[:, :ltot])
            self._tap(key + ".v", vtj.view(self.hidden, bsz, lpad)[:, :, :ltot].permute(1, 2, 0))

    def _mlp_embed(self, x, k):
        h = ops.linear(x, *self.w[k + ".in_layer"])
        h = ops.silu(h, out=h)
        return ops.linear(h, *self.w[k + ".out_layer"])

    def _mod(self, mods, key, n):
        """n chunks [B, hidden] (views into the one modulation GEMM output; common row stride)."""
        o = self._mod_off[key]
        hs = self.hidden
        return [mods[:, o + i * hs:o + (i + 1) * hs] for i in range(n)]

    def _attend(self, qj, kj, vtj, bsz, ltot, lpad):
        hd, H, D = self.hidden, self.heads, self.head_dim
        return ops.attention(qj, kj, vtj, batch=bsz, heads=H, nq=ltot, nk=ltot, nk_pad=lpad, dpad=D, scale=D ** -0.5,
                             q_bs=lpad * hd, q_rs=hd, k_bs=lpad * hd, k_rs=hd, vt_bs=lpad, vt_hs=D * bsz * lpad, vt_ds=bsz * lpad)

    def forward(self, x, timestep, context, y, guidance=None, **kwargs):
        """x [B,16,h,w], timestep [B] (sigma), context [B,Lt,ctx_dim], y [B,vec], guidance [B] -> [B,16,h,w] (x.dtype)."""
        dev, hs, H, D = self.device, self.hidden, self.heads, self.head_dim
        bsz, c, h, w = x.shape
        pad_h, pad_w = (2 - h % 2) % 2, (2 - w % 2) % 2
        xf = x.to(device=dev, dtype=self.dtype)
        if pad_h or pad_w:
            xf = torch.nn.functional.pad(xf, (0, pad_w, 0, pad_h), mode="circular")
        h_len, w_len = xf.shape[-2] // 2, xf.shape[-1] // 2
        L = h_len * w_len
        img_tok = xf.view(bsz, c, h_len, 2, w_len, 2).permute(0, 2, 4, 1, 3, 5).reshape(bsz * L, c * 4).contiguous()  # flux.py:406
        ctx = context.to(device=dev, dtype=self.dtype).contiguous()
        lt = ctx.shape[1]
        ltot = lt + L
        lpad = -(-ltot // 64) * 64
        pe = self._pe(h_len, w_len, lt)

        # ---- vec (flux.py:375-381) and every adaLN modulation of the network in one GEMM -------------------------
        t32 = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        vec = self._mlp_embed(ops.timestep_embedding(t32 * 1000.0, 256, dtype=self.dtype), "time_in")
        if self.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            g32 = guidance.to(device=dev, dtype=torch.float32).reshape(-1)
            vec = ops.linear(ops.silu(ops.linear(ops.timestep_embedding(g32 * 1000.0, 256, dtype=self.dtype), *self.w["guidance_in.in_layer"])),
                             *self.w["guidance_in.out_layer"], residual=vec)
        yv = y.to(device=dev, dtype=self.dtype).contiguous()
        vec = ops.linear(ops.silu(ops.linear(yv, *self.w["vector_in.in_layer"])), *self.w["vector_in.out_layer"], residual=vec)
        mods = ops.linear(ops.silu(vec), *self.w["mods"])                       # [B, total]
        self._tap("vec", vec)

        img = ops.linear(img_tok, *self.w["img_in"])                            # [B*L, hs]
        txt = ops.linear(ctx.view(bsz * lt, -1), *self.w["txt_in"])             # [B*Lt, hs]
        self._tap("img_in", img.view(bsz, L, hs))
        self._tap("txt_in", txt.view(bsz, lt, hs))
        qj = torch.zeros(bsz, lpad, hs, dtype=self.dtype, device=dev)           # joint txt||img q, k (pad rows stay 0)
        kj = torch.zeros(bsz, lpad, hs, dtype=self.dtype, device=dev)
        vtj = torch.zeros(hs, bsz * lpad, dtype=self.dtype, device=dev)         # V^T

        # ---- double-stream blocks (flux.py:206-264) ----------------------------------------------------------------
        streams = (("img", L, lt), ("txt", lt, 0))
        for i in range(self.depth):
            b = f"double_blocks.{i}"
            cur = {"img": img, "txt": txt}
            md = {}
            for st, n_tok, row_off in streams:
                md[st] = self._mod(mods, f"{b}.{st}", 6)
                xm = ops.layernorm_mod(cur[st], md[st][1], md[st][0], n_tok)
                qkv = ops.linear(xm, *self.w[f"{b}.{st}.qkv"])
                ops.flux_qk_norm_rope(qkv, self.w[f"{b}.{st}.qs"], self.w[f"{b}.{st}.ks"], pe, qj, kj, vtj, batch=bsz, tokens=n_tok,
                                      heads=H, head_dim=D, row_off=row_off, l_pad=lpad)
            self._tap_qkv(b, qj, kj, vtj, bsz, ltot, lpad)
            attn = self._attend(qj, kj, vtj, bsz, ltot, lpad).view(bsz, ltot, hs)
            self._tap(b + ".attn", attn)
            for st, n_tok, row_off in streams:
                xs = cur[st]
                pw, pb = self.w[f"{b}.{st}.proj"]
                for bi in range(bsz):   # the stream's rows of the joint attention output are contiguous per batch element
                    rows = xs[bi * n_tok:(bi + 1) * n_tok]
                    ops.conv_gemm(attn[bi, row_off:row_off + n_tok], pw, hs, bias=pb, gate=md[st][2][bi:bi + 1], residual=rows, out=rows, ld_out=hs)
                self._tap(f"{b}.{st}.a", xs.view(bsz, n_tok, hs))
                xm = ops.layernorm_mod(xs, md[st][4], md[st][3], n_tok)
                hdn = ops.conv_gemm(xm, self.w[f"{b}.{st}.mlp0"][0], self.mlp, bias=self.w[f"{b}.{st}.mlp0"][1], act=ops.ACT_GELU_TANH)
                self._tap(f"{b}.{st}.h", hdn.view(bsz, n_tok, self.mlp))
                ops.conv_gemm(hdn, self.w[f"{b}.{st}.mlp2"][0], hs, n=bsz, h=1, w=n_tok, bias=self.w[f"{b}.{st}.mlp2"][1], gate=md[st][5],
                              residual=xs, out=xs, ld_out=hs)
                self._tap(f"{b}.{st}", xs.view(bsz, n_tok, hs))

        # ---- single-stream blocks on the joint sequence (flux.py:283-307) ------------------------------------------
        xj = torch.cat((txt.view(bsz, lt, hs), img.view(bsz, L, hs)), 1).reshape(bsz * ltot, hs).contiguous()  # flux.py:392
        for i in range(self.depth_single):
            b = f"single_blocks.{i}"
            shift, scale, gate = self._mod(mods, b, 3)
            xm = ops.layernorm_mod(xj, scale, shift, ltot)
            qkv = ops.linear(xm, *self.w[b + ".qkv"])
            mlp = ops.conv_gemm(xm, self.w[b + ".mlp"][0], self.mlp, bias=self.w[b + ".mlp"][1], act=ops.ACT_GELU_TANH)
            ops.flux_qk_norm_rope(qkv, self.w[b + ".qs"], self.w[b + ".ks"], pe, qj, kj, vtj, batch=bsz, tokens=ltot, heads=H, head_dim=D,
                                  row_off=0, l_pad=lpad)
            self._tap_qkv(b, qj, kj, vtj, bsz, ltot, lpad)
            self._tap(b + ".mlp", mlp.view(bsz, ltot, self.mlp))
            attn = self._attend(qj, kj, vtj, bsz, ltot, lpad)
            self._tap(b + ".attn", attn.view(bsz, ltot, hs))
            ops.conv_gemm(attn, self.w[b + ".lin2"][0], hs, x1=mlp, n=bsz, h=1, w=ltot, bias=self.w[b + ".lin2"][1], gate=gate, residual=xj,
                          out=xj, ld_out=hs)
            self._tap(b, xj.view(bsz, ltot, hs))

        # ---- final layer (flux.py:317-328) + unpatchify (:416) -------------------------------------------------------
        imgf = xj.view(bsz, ltot, hs)[:, lt:].reshape(bsz * L, hs).contiguous()
        shift, scale = self._mod(mods, "final", 2)
        out = ops.linear(ops.layernorm_mod(imgf, scale, shift, L), *self.w["final_layer.linear"])     # [B*L, 64]
        out = out.view(bsz, h_len, w_len, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(bsz, c, h_len * 2, w_len * 2)
        self._tap("out", out[:, :, :h, :w])
        return out[:, :, :h, :w].to(x.dtype)

    __call__ = forward
