"""MI355X-native CLIP text encoder -- drop-in for `IntegratedCLIP` (backend/nn/clip.py:4-12), i.e. transformers'
CLIPTextModel (+ text_projection) as the classic text-processing engine drives it (classic_engine.py:124-148).

Same kernels as the UNet's transformer blocks: tokens are a [B*77, C] fp16 matrix; Q|K run as one GEMM and V is produced
transposed by the operand-swapped GEMM, the fused attention kernel gets the causal flag (key j > query i masked), LayerNorm
is the one-pass kernel, the MLP activation (quick_gelu for CLIP-L, erf-GELU for bigG) a small elementwise kernel.  The text
encoder runs once per job (its output is what gets broadcast to the other GPUs), so nothing here is tuned for throughput.
Checkpoint keys are the reference's (`transformer.text_model.*`, `transformer.text_projection.weight`)."""
import torch

from ... import hipops as ops

P = "transformer.text_model."


class IntegratedCLIP:
    def __init__(self, config, state_dict, device="cuda"):
        self.config = dict(config)
        self.device = torch.device(device)
        self.hidden = config["hidden_size"]
        self.heads = config["num_attention_heads"]
        self.layers = config["num_hidden_layers"]
        self.d = self.hidden // self.heads
        if self.d != 64:
            raise NotImplementedError("CLIP head width must be 64 (CLIP-L, OpenCLIP-H/bigG)")
        if self.hidden % 64 or config["intermediate_size"] % 64:
            raise NotImplementedError("hidden / intermediate sizes must be multiples of 64")
        self.act = ops.ACT_QUICK_GELU if config.get("hidden_act", "quick_gelu") == "quick_gelu" else ops.ACT_GELU_ERF
        self._load(state_dict)

    def _load(self, sd):
        dev = self.device

        def T(k):
            return sd[k].to(device=dev, dtype=torch.float16).contiguous()

        w = {"tok": T(P + "embeddings.token_embedding.weight"), "pos": T(P + "embeddings.position_embedding.weight"),
             "final": (T(P + "final_layer_norm.weight"), T(P + "final_layer_norm.bias"))}
        for i in range(self.layers):
            k = f"{P}encoder.layers.{i}."
            w[f"{i}.ln1"] = (T(k + "layer_norm1.weight"), T(k + "layer_norm1.bias"))
            w[f"{i}.ln2"] = (T(k + "layer_norm2.weight"), T(k + "layer_norm2.bias"))
            w[f"{i}.qk"] = (torch.cat([T(k + "self_attn.q_proj.weight"), T(k + "self_attn.k_proj.weight")], 0).contiguous(),
                            torch.cat([T(k + "self_attn.q_proj.bias"), T(k + "self_attn.k_proj.bias")], 0).contiguous())
            w[f"{i}.v"] = T(k + "self_attn.v_proj.weight")
            # softmax rows sum to 1, so V's bias passes through attention unchanged: out_proj(o + b_v) = out_proj(o) + W_o b_v.
            # Folded into out_proj's bias once at load time (fp32), which also keeps V^T = W_v X^T a bias-free swapped GEMM.
            wo, bo = sd[k + "self_attn.out_proj.weight"].float(), sd[k + "self_attn.out_proj.bias"].float()
            bo = bo + wo @ sd[k + "self_attn.v_proj.bias"].float()
            w[f"{i}.out"] = (T(k + "self_attn.out_proj.weight"), bo.to(device=dev, dtype=torch.float16).contiguous())
            w[f"{i}.fc1"] = (T(k + "mlp.fc1.weight"), T(k + "mlp.fc1.bias"))
            w[f"{i}.fc2"] = (T(k + "mlp.fc2.weight"), T(k + "mlp.fc2.bias"))
        self.projection = T("transformer.text_projection.weight") if "transformer.text_projection.weight" in sd else None
        self.w = w
        torch.cuda.synchronize(dev)

    @torch.inference_mode()
    def hidden_states(self, ids, fixes=None):
        """ids [B, T] (any integer dtype) -> list of fp16 [B*T, C]: embeddings, after layer 1, ..., after layer N (pre final LN),
        i.e. transformers' `output_hidden_states`.  fixes: per prompt [(offset, vectors [n, C]), ...] textual-inversion embeddings that
        replace the TOKEN embeddings at positions offset+1... (classic_engine.py:20-50); the position embedding still applies."""
        b, t = ids.shape
        if t > self.w["pos"].shape[0]:
            raise ValueError(f"{t} tokens > max_position_embeddings {self.w['pos'].shape[0]}")
        c, H, d = self.hidden, self.heads, self.d
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        x = ops.embed_tokens(ids32, self.w["tok"], self.w["pos"])
        if fixes is not None:
            xv = x.view(b, t, c)
            for bi, fx in enumerate(fixes):
                for offset, emb in fx:
                    n = min(t - offset - 1, emb.shape[0])
                    if n > 0:
                        xv[bi, offset + 1:offset + 1 + n] = emb[:n].to(device=self.device, dtype=torch.float16) + self.w["pos"][offset + 1:offset + 1 + n]
        tp = -(-t // 64) * 64  # keys padded to the attention tile; the padding rows/columns are zeros and masked by nk
        hs = [x]
        for i in range(self.layers):
            h = ops.layernorm(x, *self.w[f"{i}.ln1"])
            qk = torch.zeros(b, tp, 2 * c, dtype=torch.float16, device=self.device)
            vt = torch.zeros(c, b * tp, dtype=torch.float16, device=self.device)
            hv = h.view(b, t, c)
            for bi in range(b):
                ops.linear(hv[bi], *self.w[f"{i}.qk"], out=qk[bi, :t], ld_out=2 * c)
                ops.conv_gemm(self.w[f"{i}.v"], hv[bi], t, out=vt[:, bi * tp:bi * tp + t], ld_out=b * tp)   # V^T = W_v X^T
            o = ops.attention(qk, qk[:, :, c:], vt, batch=b, heads=H, nq=t, nk=t, nk_pad=tp, dpad=d, scale=d ** -0.5,
                              q_bs=tp * 2 * c, q_rs=2 * c, k_bs=tp * 2 * c, k_rs=2 * c, vt_bs=tp, vt_hs=d * b * tp, vt_ds=b * tp, causal=True)
            x = ops.linear(o, *self.w[f"{i}.out"], residual=x)
            h = ops.layernorm(x, *self.w[f"{i}.ln2"])
            h = ops.act(ops.linear(h, *self.w[f"{i}.fc1"]), self.act)
            x = ops.linear(h, *self.w[f"{i}.fc2"], residual=x)
            hs.append(x)
        return hs

    def final_layer_norm(self, x):
        return ops.layernorm(x.contiguous(), *self.w["final"])

    @torch.inference_mode()
    def encode(self, ids, clip_skip=1, final_layer_norm=True, return_pooled=False, project_pooled=False, fixes=None):
        """classic_engine.py:124-148 -> (z fp32 [B, T, C], pooled fp32 [B, C] or None)"""
        b, t = ids.shape
        hs = self.hidden_states(ids, fixes)
        z = hs[-clip_skip]
        if final_layer_norm:
            z = self.final_layer_norm(z)
        pooled = None
        if return_pooled:
            last = self.final_layer_norm(hs[-1]).view(b, t, -1)
            eos = ids.to(self.device).argmax(dim=-1)  # eos_token_id == 2 legacy branch: the EOS token has the largest id
            pooled = last[torch.arange(b, device=self.device), eos].contiguous()
            if project_pooled and self.projection is not None:
                pooled = ops.linear(pooled, self.projection)
            pooled = pooled.float()
        return z.view(b, t, -1).float(), pooled
