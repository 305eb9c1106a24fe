"""MI355X-native T5 encoder -- drop-in for `IntegratedT5` (backend/nn/t5.py:196-214), the T5-XXL text encoder whose [B, 256, 4096] output is Flux's
`txt` (backend/diffusion_engine/flux.py:87-88, backend/text_processing/t5_engine.py:59-66).

Same kernels as the other transformer stacks: tokens are a [B*T, C] matrix in fp16 or bf16; RMS LayerNorm is `fmx_rmsnorm` (T5LayerNorm, :15-25: no mean,
no bias); Q|K run as one GEMM and V is produced transposed by the operand-swapped GEMM; the fused attention kernel runs with scale 1 (the reference
multiplies K by sqrt(d) to cancel SDPA's 1/sqrt(d), :137) and takes the bucketed relative-position bias -- computed ONCE per sequence length from block 0's
32 x heads table and shared by every block (:112-125, :139-146, :190-193) -- through its additive-mask operand; the gated feed-forward
gelu_tanh(wi_0 x) * (wi_1 x) (:41-52) is the wi_1 GEMM followed by the wi_0 GEMM with the tanh-GELU epilogue and wi_1's result as its per-row gate.
Like the reference's engine, no padding mask is passed: all 256 positions attend.  The encoder runs once per prompt, nothing here is tuned for throughput.

bf16 is the type to load real T5-XXL weights in (its residual stream leaves fp16's range, as it does in the reference's fp16 run); fp16 is kept for parity
work on random-init weights.  Checkpoint keys are the reference's (`transformer.shared.weight`, `transformer.encoder.block.N.layer.{0,1}. ...`)."""
import math

import torch

from ... import hipops as ops

P = "transformer.encoder.block."


class _Transformer:
    """what `T5TextProcessingEngine` calls: text_encoder.transformer(input_ids=tokens) (t5_engine.py:22, :59-66)"""

    def __init__(self, owner):
        self._owner = owner
        self.shared = owner   # the engine moves `text_encoder.shared` to the device in fp32 (t5_engine.py:62); a no-op here

    def __call__(self, input_ids, *args, **kwargs):
        if kwargs.get("attention_mask") is not None or args:
            raise NotImplementedError("the native T5 encoder takes no attention mask (the reference's engine passes none)")
        return self._owner.encode(input_ids)


class IntegratedT5:
    def __init__(self, config, state_dict, device="cuda", dtype=torch.bfloat16):
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError(f"T5 element type {dtype}")
        self.config = dict(config)
        self.device = torch.device(device)
        self.dtype = dtype
        self.c, self.ff, self.layers, self.heads = config["d_model"], config["d_ff"], config["num_layers"], config["num_heads"]
        self.d = self.c // self.heads          # the reference builds the attention at inner_dim = d_model (t5.py:186)
        if self.d != 64 or self.c % 64 or self.ff % 64:
            raise NotImplementedError("T5 head width must be 64 and d_model / d_ff multiples of 64 (T5-XXL: 64 heads of 64, d_ff 10240)")
        if config.get("dense_act_fn", "gelu_pytorch_tanh") not in ("gelu_pytorch_tanh", "gelu_new") or not config.get("is_gated_act", True):
            raise NotImplementedError("only the gated tanh-GELU feed-forward of T5 v1.1 / T5-XXL (dense_act_fn gelu_pytorch_tanh | gelu_new, is_gated_act)")
        if config.get("model_type") == "umt5":
            raise NotImplementedError("umt5 (a relative-position table per block)")
        self.transformer = _Transformer(self)
        self.tap = None    # test hook: a list that receives the stream after the embedding, after every block, and the final norm's output
        self._bias = {}
        self._load(state_dict)

    def to(self, *a, **k):     # (`text_encoder.shared.to(device=..., dtype=float32)` of the reference's engine)
        return self

    def _load(self, sd):
        dev, dt = self.device, self.dtype

        def T(k):
            return sd[k].to(device=dev, dtype=dt).contiguous()
        w = {"tok": T("transformer.shared.weight"), "final": T("transformer.encoder.final_layer_norm.weight"),
             "zero_pos": None,
             "bias_table": sd[P + "0.layer.0.SelfAttention.relative_attention_bias.weight"].to(device=dev, dtype=torch.float32)}
        for i in range(self.layers):
            a, f = f"{P}{i}.layer.0.", f"{P}{i}.layer.1."
            w[f"{i}.ln0"], w[f"{i}.ln1"] = T(a + "layer_norm.weight"), T(f + "layer_norm.weight")
            w[f"{i}.qk"] = torch.cat([T(a + "SelfAttention.q.weight"), T(a + "SelfAttention.k.weight")], 0).contiguous()
            w[f"{i}.v"], w[f"{i}.o"] = T(a + "SelfAttention.v.weight"), T(a + "SelfAttention.o.weight")
            w[f"{i}.wi0"], w[f"{i}.wi1"], w[f"{i}.wo"] = T(f + "DenseReluDense.wi_0.weight"), T(f + "DenseReluDense.wi_1.weight"), T(f + "DenseReluDense.wo.weight")
        self.w = w
        torch.cuda.synchronize(dev)

    @staticmethod
    def _relative_position_bucket(rel, num_buckets=32, max_distance=128):
        """t5.py:88-110 (bidirectional): host integer arithmetic, bit-exact with the reference's"""
        nb = num_buckets // 2
        out = (rel > 0).to(torch.long) * nb
        rel = torch.abs(rel)
        max_exact = nb // 2
        large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
        large = torch.min(large, torch.full_like(large, nb - 1))
        return out + torch.where(rel < max_exact, rel, large)

    def position_bias(self, t, tp):
        """additive bias [H, t, tp] in the element type (keys beyond t are masked by nk): once per sequence length (t5.py:112-125)"""
        key = (t, tp)
        if key not in self._bias:
            pos = torch.arange(t, dtype=torch.long)
            buckets = self._relative_position_bucket(pos[None, :] - pos[:, None]).to(self.device)
            vals = self.w["bias_table"][buckets].permute(2, 0, 1)             # [H, t, t]
            b = torch.zeros(self.heads, t, tp, dtype=self.dtype, device=self.device)
            b[:, :, :t] = vals.to(self.dtype)
            self._bias[key] = b.contiguous()
        return self._bias[key]

    @torch.inference_mode()
    def encode(self, ids):
        """ids [B, T] -> fp32 [B, T, d_model] (T5.forward, t5.py:204-209)"""
        b, t = ids.shape
        c, H, d = self.c, self.heads, self.d
        m = b * t
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        x = self.w["tok"][ids32.long()].reshape(m, c).contiguous()              # nn.Embedding gather (:205); no position embedding in T5
        tp = -(-t // 64) * 64
        bias = self.position_bias(t, tp)
        if self.tap is not None:
            self.tap.append(x.float().cpu().view(b, t, c))
        for i in range(self.layers):
            n = ops.rmsnorm(x, self.w[f"{i}.ln0"], 1e-6)
            if tp == t:
                qk = ops.linear(n, self.w[f"{i}.qk"])                            # [B*T, 2C] = [Q | K]
                vt = ops.conv_gemm(self.w[f"{i}.v"], n, m)                       # [C, B*T] = V^T (operand swap)
            else:                                                                # keys padded to the attention tile per prompt; pad rows are zeros, masked by nk
                qk = torch.zeros(b, tp, 2 * c, dtype=self.dtype, device=self.device)
                vt = torch.zeros(c, b * tp, dtype=self.dtype, device=self.device)
                nv = n.view(b, t, c)
                for bi in range(b):
                    ops.linear(nv[bi], self.w[f"{i}.qk"], out=qk[bi, :t], ld_out=2 * c)
                    ops.conv_gemm(self.w[f"{i}.v"], nv[bi], t, out=vt[:, bi * tp:bi * tp + t], ld_out=b * tp)
            qk3 = qk.view(b, tp, 2 * c)
            o = ops.attention(qk3, qk3[:, :, c:], vt, batch=b, heads=H, nq=t, nk=t, nk_pad=tp, dpad=d, scale=1.0, q_bs=tp * 2 * c, q_rs=2 * c,
                              k_bs=tp * 2 * c, k_rs=2 * c, vt_bs=tp, vt_hs=d * b * tp, vt_ds=b * tp, mask=bias, mask_strides=(0, t * tp, tp))
            x = ops.linear(o, self.w[f"{i}.o"], residual=x)
            n = ops.rmsnorm(x, self.w[f"{i}.ln1"], 1e-6)
            lin = ops.linear(n, self.w[f"{i}.wi1"])                              # hidden_linear = wi_1 x
            h = ops.conv_gemm(n, self.w[f"{i}.wi0"], self.ff, act=ops.ACT_GELU_TANH, gate=lin, n=m, h=1, w=1)   # gelu_tanh(wi_0 x) * hidden_linear, per row
            x = ops.linear(h, self.w[f"{i}.wo"], residual=x)
            if self.tap is not None:
                self.tap.append(x.float().cpu().view(b, t, c))
        y = ops.rmsnorm(x, self.w["final"], 1e-6).view(b, t, c).float()
        if self.tap is not None:
            self.tap.append(y.cpu())
        return y

    __call__ = encode
