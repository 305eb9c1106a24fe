"""diffusers <-> LDM parameter names of the UNet -- counterpart of backend/misc/diffusers_state_dict.py:70
(`unet_to_diffusers`, also packages_3rdparty/comfyui_lora_collection/utils.py:189), which LoRA files in the diffusers /
kohya naming need (`lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q`, ...).

Built from the executor's own flat layout (`layout.unet_layout`) rather than from the config lists: every layer object
already knows its LDM key, so the diffusers name only needs the (level, index-within-level) bookkeeping of diffusers'
down_blocks / mid_block / up_blocks.  tests/test_loader_lora.py checks the result against the reference's map."""
import torch

from ..nn.layout import Down, Res, SpatialT, Up, unet_layout, unet_param_shapes

_RES = {  # LDM ResBlock sub-key -> diffusers ResnetBlock2D sub-key
    "in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj",
    "out_layers.0": "norm2", "out_layers.3": "conv2", "skip_connection": "conv_shortcut",
}
_BASIC = {  # diffusers -> LDM (both embedding names of diffusers map to label_emb)
    "class_embedding.linear_1": "label_emb.0.0", "class_embedding.linear_2": "label_emb.0.2",
    "add_embedding.linear_1": "label_emb.0.0", "add_embedding.linear_2": "label_emb.0.2",
    "conv_in": "input_blocks.0.0", "conv_norm_out": "out.0", "conv_out": "out.2",
    "time_embedding.linear_1": "time_embed.0", "time_embedding.linear_2": "time_embed.2",
}


def unet_to_diffusers(unet_config):
    """-> {diffusers parameter name: LDM parameter name} for every parameter both namings have."""
    if "num_res_blocks" not in unet_config:
        return {}
    lay = unet_layout(unet_config)
    have = set(unet_param_shapes(unet_config))
    out = {}

    def put(dname, lname):
        for suf in (".weight", ".bias"):
            # the reference's table lists the skip / embedding entries unconditionally; keep them all (harmless extra keys)
            out[dname + suf] = lname + suf

    def res(dprefix, L):
        for lk, dk in _RES.items():
            put(f"{dprefix}.{dk}", f"{L.key}.{lk}")

    def attn(dprefix, L):
        for n in ("proj_in", "proj_out", "norm"):
            put(f"{dprefix}.{n}", f"{L.key}.{n}")
        for t in range(L.depth):
            for sub in ("norm1", "norm2", "norm3", "attn1.to_out.0", "attn2.to_out.0", "ff.net.0.proj", "ff.net.2"):
                put(f"{dprefix}.transformer_blocks.{t}.{sub}", f"{L.key}.transformer_blocks.{t}.{sub}")
            for a in ("attn1", "attn2"):
                for q in ("to_q", "to_k", "to_v"):
                    out[f"{dprefix}.transformer_blocks.{t}.{a}.{q}.weight"] = f"{L.key}.transformer_blocks.{t}.{a}.{q}.weight"

    # down path: diffusers level x = number of Downsample layers seen so far, i = index of the ResBlock inside the level
    level, i = 0, 0
    for blk in lay.input_blocks[1:]:
        if isinstance(blk[0], Down):
            put(f"down_blocks.{level}.downsamplers.0.conv", blk[0].key + ".op")
            level, i = level + 1, 0
            continue
        for L in blk:
            if isinstance(L, Res):
                res(f"down_blocks.{level}.resnets.{i}", L)
            elif isinstance(L, SpatialT):
                attn(f"down_blocks.{level}.attentions.{i}", L)
        i += 1
    # the reference emits a downsampler entry for the last level too (it does not exist in the model; harmless)
    nrb = unet_config["num_res_blocks"]
    nrb = [nrb] * len(unet_config.get("channel_mult", (1, 2, 4, 8))) if isinstance(nrb, int) else list(nrb)
    n_last = sum(r + 1 for r in nrb)  # = 1 + blocks of the earlier levels + the last level's ResBlocks
    put(f"down_blocks.{len(nrb) - 1}.downsamplers.0.conv", f"input_blocks.{n_last}.0.op")
    # middle
    ri = 0
    for L in lay.middle:
        if isinstance(L, Res):
            res(f"mid_block.resnets.{ri}", L)
            ri += 1
        else:
            attn("mid_block.attentions.0", L)
    # up path
    level, i = 0, 0
    for blk in lay.output_blocks:
        for L in blk:
            if isinstance(L, Res):
                res(f"up_blocks.{level}.resnets.{i}", L)
            elif isinstance(L, SpatialT):
                attn(f"up_blocks.{level}.attentions.{i}", L)
            elif isinstance(L, Up):
                put(f"up_blocks.{level}.upsamplers.0.conv", L.key + ".conv")
        i += 1
        if i == nrb[::-1][level] + 1:
            if not any(isinstance(L, Up) for L in blk):  # last level: the reference still lists an upsampler slot
                c = 1 + sum(isinstance(L, SpatialT) for L in blk)
                put(f"up_blocks.{level}.upsamplers.0.conv", f"{blk[0].key[:-2]}.{c}.conv")
            level, i = level + 1, 0
    for dk, lk in _BASIC.items():
        put(dk, lk)
    return out


def vae_from_diffusers(sd):
    """diffusers `AutoencoderKL` parameter names -> the LDM names the native VAE loads (the reference calls
    huggingface_guess.diffusers_convert.convert_vae_state_dict for this, backend/loader.py:58-59; that package is not vendored, the mapping
    is the published one of diffusers' own conversion script):
      {en,de}coder.{down,up}_blocks.i.resnets.j -> encoder.down.i.block.j / decoder.up.(n-1-i).block.j   (the decoder counts levels backwards)
      downsamplers.0.conv -> downsample.conv, upsamplers.0.conv -> upsample.conv, mid_block.resnets.j -> mid.block_(j+1),
      mid_block.attentions.0.{group_norm, to_q, to_k, to_v, to_out.0} -> mid.attn_1.{norm, q, k, v, proj_out} (Linear [C, C] -> 1x1 conv),
      conv_shortcut -> nin_shortcut, conv_norm_out -> norm_out.
    A dict that is already LDM-keyed is returned unchanged."""
    import re
    if not any(".up_blocks." in k or ".down_blocks." in k or ".mid_block." in k for k in sd):
        return sd
    n_up = 1 + max(int(m.group(1)) for m in (re.match(r"decoder\.up_blocks\.(\d+)\.", k) for k in sd) if m)
    attn = {"group_norm": "norm", "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out",
            "query": "q", "key": "k", "value": "v", "proj_attn": "proj_out"}   # second row: names of diffusers < 0.15
    out = {}
    for k, v in sd.items():
        m = re.match(r"(encoder|decoder)\.(down_blocks|up_blocks)\.(\d+)\.(resnets\.(\d+)|downsamplers\.0\.conv|upsamplers\.0\.conv)\.(.+)", k)
        if m:
            side, kind, i, what, j, rest = m.group(1), m.group(2), int(m.group(3)), m.group(4), m.group(5), m.group(6)
            level = i if kind == "down_blocks" else n_up - 1 - i
            stem = f"{side}.{'down' if kind == 'down_blocks' else 'up'}.{level}"
            if what.startswith("resnets"):
                k = f"{stem}.block.{j}.{rest}"
            else:
                k = f"{stem}.{'downsample' if what.startswith('down') else 'upsample'}.conv.{rest}"
        else:
            m = re.match(r"(encoder|decoder)\.mid_block\.resnets\.(\d+)\.(.+)", k)
            if m:
                k = f"{m.group(1)}.mid.block_{int(m.group(2)) + 1}.{m.group(3)}"
            else:
                m = re.match(r"(encoder|decoder)\.mid_block\.attentions\.0\.(.+)\.(weight|bias)", k)
                if m:
                    k = f"{m.group(1)}.mid.attn_1.{attn[m.group(2)]}.{m.group(3)}"
                    if m.group(3) == "weight" and v.dim() == 2:
                        v = v.reshape(v.shape[0], v.shape[1], 1, 1)
        k = k.replace(".conv_shortcut.", ".nin_shortcut.").replace(".conv_norm_out.", ".norm_out.")
        out[k] = v
    return out


# ---- Flux: diffusers transformer names -> the single-file (BFL) names the native transformer is keyed by ------------------------------------------------
def swap_scale_shift(weight):
    """diffusers stores the final adaLN projection as [scale | shift], the BFL layout as [shift | scale] (comfyui_lora_collection/utils.py:255-258)"""
    shift, scale = weight.chunk(2, dim=0)
    return torch.cat([scale, shift], dim=0)


def flux_to_diffusers(cfg, output_prefix=""):
    """{diffusers parameter name: target} for a Flux transformer (depth, depth_single_blocks, hidden_size), as
    packages_3rdparty/comfyui_lora_collection/utils.py:407-509 builds it.  A target is the BFL parameter name, or (name, (dim, offset, size)) where the
    diffusers tensor is one slice of a fused projection (q | k | v of `img_attn.qkv` / `txt_attn.qkv`; q | k | v | mlp of a single block's `linear1`), or
    (name, None, function) where it maps through a function (`swap_scale_shift`)."""
    hs = cfg.get("hidden_size", 0)
    out = {}

    def fused(dst, names_sizes, src_prefix):
        for end in ("weight", "bias"):
            off = 0
            for name, size in names_sizes:
                out[f"{src_prefix}{name}.{end}"] = (f"{dst}.{end}", (0, off, size))
                off += size
    double = [("attn.to_out.0", "img_attn.proj"), ("norm1.linear", "img_mod.lin"), ("norm1_context.linear", "txt_mod.lin"), ("attn.to_add_out", "txt_attn.proj"),
              ("ff.net.0.proj", "img_mlp.0"), ("ff.net.2", "img_mlp.2"), ("ff_context.net.0.proj", "txt_mlp.0"), ("ff_context.net.2", "txt_mlp.2")]
    double_scales = [("attn.norm_q.weight", "img_attn.norm.query_norm.scale"), ("attn.norm_k.weight", "img_attn.norm.key_norm.scale"),
                     ("attn.norm_added_q.weight", "txt_attn.norm.query_norm.scale"), ("attn.norm_added_k.weight", "txt_attn.norm.key_norm.scale")]
    for i in range(cfg.get("depth", 0)):
        src, dst = f"transformer_blocks.{i}.", f"{output_prefix}double_blocks.{i}."
        fused(dst + "img_attn.qkv", [("to_q", hs), ("to_k", hs), ("to_v", hs)], src + "attn.")
        fused(dst + "txt_attn.qkv", [("add_q_proj", hs), ("add_k_proj", hs), ("add_v_proj", hs)], src + "attn.")
        for a, b in double:
            for end in ("weight", "bias"):
                out[f"{src}{a}.{end}"] = f"{dst}{b}.{end}"
        for a, b in double_scales:
            out[src + a] = dst + b
    for i in range(cfg.get("depth_single_blocks", 0)):
        src, dst = f"single_transformer_blocks.{i}.", f"{output_prefix}single_blocks.{i}."
        for end in ("weight", "bias"):
            off = 0
            for name, size in (("attn.to_q", hs), ("attn.to_k", hs), ("attn.to_v", hs), ("proj_mlp", 4 * hs)):
                out[f"{src}{name}.{end}"] = (f"{dst}linear1.{end}", (0, off, size))
                off += size
            out[f"{src}norm.linear.{end}"] = f"{dst}modulation.lin.{end}"
            out[f"{src}proj_out.{end}"] = f"{dst}linear2.{end}"
        out[src + "attn.norm_q.weight"] = dst + "norm.query_norm.scale"
        out[src + "attn.norm_k.weight"] = dst + "norm.key_norm.scale"
    basic = [("final_layer.linear", "proj_out"), ("img_in", "x_embedder"), ("time_in.in_layer", "time_text_embed.timestep_embedder.linear_1"),
             ("time_in.out_layer", "time_text_embed.timestep_embedder.linear_2"), ("txt_in", "context_embedder"),
             ("vector_in.in_layer", "time_text_embed.text_embedder.linear_1"), ("vector_in.out_layer", "time_text_embed.text_embedder.linear_2"),
             ("guidance_in.in_layer", "time_text_embed.guidance_embedder.linear_1"), ("guidance_in.out_layer", "time_text_embed.guidance_embedder.linear_2")]
    for bfl, dif in basic:
        for end in ("weight", "bias"):
            out[f"{dif}.{end}"] = f"{output_prefix}{bfl}.{end}"
    for end in ("weight", "bias"):
        out[f"norm_out.linear.{end}"] = (f"{output_prefix}final_layer.adaLN_modulation.1.{end}", None, swap_scale_shift)
    return out
