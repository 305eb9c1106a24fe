"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the CLIP text encoder the way the reference drives it:
  backend/nn/clip.py:4-12 (IntegratedCLIP = transformers.CLIPTextModel [+ text_projection]),
  backend/text_processing/classic_engine.py:124-148 (encode_with_transformers: hidden_states[-clip_skip], optional
  final_layer_norm, pooled output [+ text_projection for every encoder but clip_l]),
  :263-316 (process_tokens: emphasis multipliers, "Original" mean restoration of emphasis.py:34-42).
The arithmetic itself is third-party: transformers.models.clip.modeling_clip (CLIPTextTransformer; pinned 4.46.1 in the
reference's requirements_versions.txt, 5.15 installed here): token + position embeddings, pre-LN blocks with causal
self-attention (scale d^-0.5), quick_gelu / gelu MLP, final LayerNorm, pooled = last_hidden_state at argmax(input_ids) (the
eos_token_id == 2 legacy branch the SD configs take).  tests/golden/tiny_clip_*.pt pins this file against that package.
"""
import math

import torch
import torch.nn.functional as F

P = "transformer.text_model."


def _ln(sd, key, x):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


@torch.no_grad()
def clip_hidden_states(sd, cfg, ids, fixes=None):
    """-> list of hidden states (embeddings, after layer 1, ..., after layer N), all pre-final-LN (HF `hidden_states`).
    fixes: per prompt a list of (offset, vectors [n, C]) textual-inversion embeddings that REPLACE the token embeddings of positions
    offset+1 ... (classic_engine.py:20-50 CLIPEmbeddingForTextualInversion), before the position embedding is added."""
    b, t = ids.shape
    c, heads = cfg["hidden_size"], cfg["num_attention_heads"]
    d = c // heads
    tok = sd[P + "embeddings.token_embedding.weight"][ids]
    if fixes is not None:
        rows = []
        for fx, tensor in zip(fixes, tok):
            for offset, emb in fx:
                n = min(tensor.shape[0] - offset - 1, emb.shape[0])
                tensor = torch.cat([tensor[0:offset + 1], emb[0:n].to(tensor), tensor[offset + 1 + n:]])
            rows.append(tensor)
        tok = torch.stack(rows)
    x = tok + sd[P + "embeddings.position_embedding.weight"][:t][None]
    mask = torch.full((t, t), float("-inf")).triu(1)
    hs = [x]
    for i in range(cfg["num_hidden_layers"]):
        k = f"{P}encoder.layers.{i}."
        h = _ln(sd, k + "layer_norm1", x)
        q = _lin(sd, k + "self_attn.q_proj", h).view(b, t, heads, d).transpose(1, 2) * d ** -0.5
        kk = _lin(sd, k + "self_attn.k_proj", h).view(b, t, heads, d).transpose(1, 2)
        v = _lin(sd, k + "self_attn.v_proj", h).view(b, t, heads, d).transpose(1, 2)
        a = torch.softmax(q @ kk.transpose(-1, -2) + mask, dim=-1) @ v
        x = x + _lin(sd, k + "self_attn.out_proj", a.transpose(1, 2).reshape(b, t, c))
        h = _lin(sd, k + "mlp.fc1", _ln(sd, k + "layer_norm2", x))
        h = h * torch.sigmoid(1.702 * h) if cfg["hidden_act"] == "quick_gelu" else F.gelu(h)
        x = x + _lin(sd, k + "mlp.fc2", h)
        hs.append(x)
    return hs


@torch.no_grad()
def encode_with_transformers(sd, cfg, ids, clip_skip=1, final_layer_norm=True, return_pooled=False, is_clip_l=True, fixes=None):
    """classic_engine.py:124-148 -> (z [B,T,C], pooled [B,C] or None)"""
    hs = clip_hidden_states(sd, cfg, ids, fixes)
    z = hs[-clip_skip]
    if final_layer_norm:
        z = _ln(sd, P + "final_layer_norm", z)
    pooled = None
    if return_pooled:
        last = _ln(sd, P + "final_layer_norm", hs[-1])
        pooled = last[torch.arange(ids.shape[0]), ids.argmax(dim=-1)]
        if "transformer.text_projection.weight" in sd and not is_clip_l:
            pooled = F.linear(pooled, sd["transformer.text_projection.weight"])
    return z, pooled


def apply_emphasis_original(z, multipliers):
    """emphasis.py:34-42 (EmphasisOriginal.after_transformers): z *= multipliers, then restore the tensor mean."""
    original_mean = z.mean()
    z = z * multipliers.reshape(multipliers.shape + (1,)).expand(z.shape)
    new_mean = z.mean()
    return z * (original_mean / new_mean)


def timestep_embedder_256(v):
    """backend/nn/unet.py:55-67 at dim 256 (diffusion_engine/sdxl.py:72 `Timestep(256)`): [cos | sin] of v * exp(-ln 1e4 * k / 128)."""
    half = 128
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = torch.tensor([float(v)])[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


@torch.no_grad()
def sdxl_conditioning(sd_l, cfg_l, sd_g, cfg_g, ids_l, ids_g, width, height, crop_left=0, crop_top=0):
    """diffusion_engine/sdxl.py:76-117 for one 77-token chunk per prompt without emphasis."""
    cond_l, _ = encode_with_transformers(sd_l, cfg_l, ids_l, clip_skip=2, final_layer_norm=False)
    cond_g, pooled = encode_with_transformers(sd_g, cfg_g, ids_g, clip_skip=2, final_layer_norm=False, return_pooled=True, is_clip_l=False)
    flat = torch.cat([timestep_embedder_256(v) for v in (height, width, crop_top, crop_left, height, width)]).flatten()[None].repeat(pooled.shape[0], 1)
    return {"crossattn": torch.cat([cond_l, cond_g], dim=2), "vector": torch.cat([pooled, flat], dim=1)}
