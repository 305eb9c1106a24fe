"""TEST INFRASTRUCTURE ONLY (CPU oracle): the CLIP text transformer with fp16 rounding at the native executor's storage sites.

The companion of oracle/unet_fp16sites.py for SURVEY row f2 (reference backend/nn/clip.py:4-12 = transformers' CLIPTextModel driven by
backend/text_processing/classic_engine.py:124-148).  The arithmetic is oracle/clip.py's (pinned to transformers through tests/golden/tiny_clip_*.pt);
`rounding=False` reproduces its hidden states bit for bit.  With rounding on, what the native executor (stable-diffusion-webui-forge_amd/backend/nn/clip.py)
stores is rounded where it stores it: fp16 parameters (token / position tables included), their sum, LayerNorm outputs, q|k with their biases, V^T WITHOUT
its bias (folded, in fp32 from the fp32 checkpoint tensors, into out_proj's bias and rounded once), the attention output (the generic masked kernel: unscaled
q, scale x log2 e on the fp32 scores, causal mask, P rounded for P V, row sum over the unrounded exponentials), out_proj + folded bias + x with one rounding,
fc1, the activation, fc2 + bias + x with one rounding.

`teacher`: the native executor's hidden states (the list `IntegratedCLIP.hidden_states` returns): layer i is then evaluated on the NATIVE output of layer
i - 1, so that the comparison is layer-wise."""
import torch
import torch.nn.functional as F

P = "transformer.text_model."
LOG2E = 1.4426950408889634


@torch.no_grad()
def clip_hidden_states(sd, cfg, ids, rounding=True, teacher=None):
    """-> list of hidden states (embeddings, after layer 1, ..., after layer N); with `teacher` every layer starts from teacher[i - 1]"""
    R = (lambda t: t.half().float()) if rounding else (lambda t: t)
    b, t = ids.shape
    c, heads = cfg["hidden_size"], cfg["num_attention_heads"]
    d = c // heads

    def ln(key, x):
        return F.layer_norm(x, (c,), R(sd[key + ".weight"]), R(sd[key + ".bias"]), 1e-5)

    def lin(key, x, bias=True):
        return F.linear(x, R(sd[key + ".weight"]), R(sd[key + ".bias"]) if bias else None)

    x = R(R(sd[P + "embeddings.token_embedding.weight"])[ids] + R(sd[P + "embeddings.position_embedding.weight"])[:t][None])
    mask = torch.full((t, t), float("-inf")).triu(1)
    hs = [x]
    for i in range(cfg["num_hidden_layers"]):
        if teacher is not None:
            x = teacher[i].float().reshape(b, t, c)
        k = f"{P}encoder.layers.{i}."
        h = R(ln(k + "layer_norm1", x))
        q = R(lin(k + "self_attn.q_proj", h)).view(b, t, heads, d).transpose(1, 2)
        kk = R(lin(k + "self_attn.k_proj", h)).view(b, t, heads, d).transpose(1, 2)
        if rounding:
            v = R(lin(k + "self_attn.v_proj", h, bias=False)).view(b, t, heads, d).transpose(1, 2)
            s = (q @ kk.transpose(-1, -2)) * (torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)) + mask
            p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
            a = R((R(p) @ v) / p.sum(dim=-1, keepdim=True))
            bo = R(sd[k + "self_attn.out_proj.bias"].float() + sd[k + "self_attn.out_proj.weight"].float() @ sd[k + "self_attn.v_proj.bias"].float())
            x = R(F.linear(a.transpose(1, 2).reshape(b, t, c), R(sd[k + "self_attn.out_proj.weight"]), bo) + x)
        else:
            v = lin(k + "self_attn.v_proj", h).view(b, t, heads, d).transpose(1, 2)
            a = torch.softmax((q * d ** -0.5) @ kk.transpose(-1, -2) + mask, dim=-1) @ v
            x = x + lin(k + "self_attn.out_proj", a.transpose(1, 2).reshape(b, t, c))
        h = R(lin(k + "mlp.fc1", R(ln(k + "layer_norm2", x))))
        h = R(h * torch.sigmoid(1.702 * h) if cfg["hidden_act"] == "quick_gelu" else F.gelu(h))
        x = R(x + lin(k + "mlp.fc2", h)) if rounding else x + lin(k + "mlp.fc2", h)
        hs.append(x)
    return hs
