"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- never imported by the product package.

Torch-fp32 restatement of the reference's UNet forward for the SD1.x / SDXL family,
  backend/nn/unet.py:696-763  IntegratedUNet2DConditionModel.forward
written as plain functions over an LDM-keyed state dict: the block structure is *discovered from
the checkpoint keys* (ResBlock has `in_layers.0`, Downsample has `op`, SpatialTransformer has `norm` +
`transformer_blocks`, Upsample has `conv`) instead of being rebuilt as nn.Modules.

Parity status: pinned against the imported reference modules in this container
(tests/test_oracle_vs_reference.py) and against committed fixtures generated from them
(tests/golden/, oracle/make_golden.py).  The reference itself ships no tests for this path.
"""
import math

import torch
import torch.nn.functional as F

from .attention import attention


def timestep_embedding(t, dim, max_period=10000):
    # unet.py:55-67 -- cat([cos, sin]), freqs in fp32
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, key, x, eps):
    return F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], eps)


def _conv(sd, key, x, stride=1, padding=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def _ln(sd, key, x):
    w = sd[key + ".weight"]
    return F.layer_norm(x, w.shape, w, sd[key + ".bias"], 1e-5)  # torch default eps (unet.py:167-175)


def resblock(sd, key, x, emb):
    # unet.py:433-478 (no updown, no scale-shift): GN(1e-5)+SiLU+conv ; + Linear(SiLU(emb)) ; GN+SiLU+conv ; + skip
    h = _conv(sd, key + ".in_layers.2", F.silu(_gn(sd, key + ".in_layers.0", x, 1e-5)))
    e = _lin(sd, key + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = _conv(sd, key + ".out_layers.3", F.silu(_gn(sd, key + ".out_layers.0", h, 1e-5)))
    if key + ".skip_connection.weight" in sd:
        x = _conv(sd, key + ".skip_connection", x, padding=0)
    return x + h


def cross_attention(sd, key, x, context, heads):
    # unet.py:145-155
    q = _lin(sd, key + ".to_q", x)
    ctx = x if context is None else context
    k = _lin(sd, key + ".to_k", ctx)
    v = _lin(sd, key + ".to_v", ctx)
    return _lin(sd, key + ".to_out.0", attention(q, k, v, heads))


def _hooked_attention(sd, key, which, n, context, to, heads):
    """One attention sub-layer with the hook points of unet.py:205-238 (attn1) / :244-274 (attn2): `<which>_patch` may replace
    (query source, context, value source); `patches_replace[<which>]` keyed (block, id, block_index) or (block, id) takes over the
    attention itself between the projections; `<which>_output_patch` post-processes."""
    patches, replace = to.get("patches", {}), to.get("patches_replace", {})
    extra = {k: v for k, v in to.items() if k not in ("patches", "patches_replace")}
    extra["n_heads"], extra["dim_head"] = heads, n.shape[-1] // heads
    value = None
    if which + "_patch" in patches:
        if context is None:
            context = n
        value = context
        for p in patches[which + "_patch"]:
            n, context, value = p(n, context, value, extra)
    block = to.get("block")
    tb = (block[0], block[1], to.get("block_index", 0)) if block is not None else None
    rep = replace.get(which, {})
    k_ = tb if tb in rep else block
    context = n if context is None else context
    value = context if value is None else value
    q, k, v = _lin(sd, key + ".to_q", n), _lin(sd, key + ".to_k", context), _lin(sd, key + ".to_v", value)
    o = rep[k_](q, k, v, extra) if k_ in rep else attention(q, k, v, heads)
    n = _lin(sd, key + ".to_out.0", o)
    for p in patches.get(which + "_output_patch", []):
        n = p(n, extra)
    return n


def transformer_block(sd, key, x, context, heads, to=None):
    # unet.py:183-279: three pre-LN residual sub-layers (+ the patch hooks when transformer_options carries any)
    if to is not None and (to.get("patches") or to.get("patches_replace")):
        x = x + _hooked_attention(sd, key + ".attn1", "attn1", _ln(sd, key + ".norm1", x), None, to, heads)
        extra = {k: v for k, v in to.items() if k not in ("patches", "patches_replace")}
        for p in to.get("patches", {}).get("middle_patch", []):
            x = p(x, extra)
        x = x + _hooked_attention(sd, key + ".attn2", "attn2", _ln(sd, key + ".norm2", x), context, to, heads)
    else:
        x = x + cross_attention(sd, key + ".attn1", _ln(sd, key + ".norm1", x), None, heads)
        x = x + cross_attention(sd, key + ".attn2", _ln(sd, key + ".norm2", x), context, heads)
    h = _lin(sd, key + ".ff.net.0.proj", _ln(sd, key + ".norm3", x))
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)  # exact erf GELU (unet.py:111)
    return x + _lin(sd, key + ".ff.net.2", h)


def spatial_transformer(sd, key, x, context, heads, to=None):
    # unet.py:308-327
    b, c, hh, ww = x.shape
    x_in = x
    x = _gn(sd, key + ".norm", x, 1e-6)
    use_linear = sd[key + ".proj_in.weight"].ndim == 2
    if not use_linear:
        x = _conv(sd, key + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
    if use_linear:
        x = _lin(sd, key + ".proj_in", x)
    d = 0
    while f"{key}.transformer_blocks.{d}.norm1.weight" in sd:
        if to is not None:
            to["block_index"] = d
        x = transformer_block(sd, f"{key}.transformer_blocks.{d}", x, context, heads, to)
        d += 1
    if use_linear:
        x = _lin(sd, key + ".proj_out", x)
    x = x.reshape(b, hh, ww, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = _conv(sd, key + ".proj_out", x, padding=0)
    return x + x_in


def _heads_for(cfg, ch):
    nhc = cfg.get("num_head_channels", -1)
    return cfg["num_heads"] if nhc == -1 else ch // nhc


def _run_block(sd, cfg, prefix, h, emb, context, output_shape=None, to=None):
    """One TimestepEmbedSequential (unet.py:74-93): sub-layers prefix.0, prefix.1, ..."""
    j = 0
    while True:
        key = f"{prefix}.{j}"
        if key + ".in_layers.0.weight" in sd:
            h = resblock(sd, key, h, emb)
        elif key + ".norm.weight" in sd:
            h = spatial_transformer(sd, key, h, context, _heads_for(cfg, h.shape[1]), to)
            if to is not None and "transformer_index" in to:
                to["transformer_index"] += 1
        elif key + ".op.weight" in sd:
            h = _conv(sd, key + ".op", h, stride=2)  # Downsample: conv3x3 stride 2 pad 1 (unet.py:367)
        elif key + ".conv.weight" in sd:
            # Upsample (unet.py:340-355): nearest to the next skip's H,W (or x2), then conv3x3
            size = [h.shape[2] * 2, h.shape[3] * 2] if output_shape is None else list(output_shape[2:])
            h = _conv(sd, key + ".conv", F.interpolate(h, size=size, mode="nearest"))
        elif key + ".weight" in sd and sd[key + ".weight"].ndim == 4:
            h = _conv(sd, key, h)  # input_blocks.0.0
        else:
            break
        j += 1
    if j == 0:
        raise KeyError(prefix)
    return h


def _apply_control(h, control, name):
    """unet.py:44-52: pop the LAST tensor of control[name] and add it (None entries are skipped)."""
    if control is not None and name in control and len(control[name]) > 0:
        ctrl = control[name].pop()
        if ctrl is not None:
            h = h + ctrl
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, y=None, control=None, transformer_options=None):
    """x [B,C,H,W] fp32, timesteps [B] (table index as float), context [B,T,D], y [B,adm] or None -> eps.
    control: {'input': [...], 'middle': [...], 'output': [...]} ControlNet residual lists (consumed from the end, unet.py:714,732,739).
    transformer_options: the hook points of unet.py:696-763 (`block_modifiers`, `patches`: input_block_patch(_after_skip),
    output_block_patch; the per-transformer-block ones are in transformer_block above)."""
    if control is not None:
        control = {k: list(v) for k, v in control.items()}
    to = transformer_options
    patches = to.get("patches", {}) if to else {}
    mods = to.get("block_modifiers", []) if to else []
    if to is not None:
        to["original_shape"], to["transformer_index"] = list(x.shape), 0

    def modify(h, when):
        for m in mods:
            h = m(h, when, to)
        return h

    def at(block):
        if to is not None:
            to["block"] = block
    mc = cfg["model_channels"]
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, mc))))
    if "label_emb.0.0.weight" in sd:
        assert y is not None and y.shape[0] == x.shape[0]
        emb = emb + _lin(sd, "label_emb.0.2", F.silu(_lin(sd, "label_emb.0.0", y)))
    hs = []
    h = x
    i = 0
    while f"input_blocks.{i}.0.weight" in sd or any(
            f"input_blocks.{i}.0.{s}" in sd for s in ("in_layers.0.weight", "op.weight")):
        at(("input", i))
        h = modify(h, "before")
        h = _run_block(sd, cfg, f"input_blocks.{i}", h, emb, context, to=to)
        h = _apply_control(h, control, "input")
        h = modify(h, "after")
        for p in patches.get("input_block_patch", []):
            h = p(h, to)
        hs.append(h)
        for p in patches.get("input_block_patch_after_skip", []):
            h = p(h, to)
        i += 1
    at(("middle", 0))
    h = modify(h, "before")
    h = _run_block(sd, cfg, "middle_block", h, emb, context, to=to)
    h = _apply_control(h, control, "middle")
    h = modify(h, "after")
    i = 0
    while f"output_blocks.{i}.0.in_layers.0.weight" in sd:
        at(("output", i))
        hsp = _apply_control(hs.pop(), control, "output")
        for p in patches.get("output_block_patch", []):
            h, hsp = p(h, hsp, to)
        h = torch.cat([h, hsp], dim=1)  # current first (unet.py:741)
        out_shape = hs[-1].shape if hs else None
        h = modify(h, "before")
        h = _run_block(sd, cfg, f"output_blocks.{i}", h, emb, context, out_shape, to=to)
        h = modify(h, "after")
        i += 1
    at(("last", 0))
    h = modify(h, "before")
    h = _conv(sd, "out.2", F.silu(_gn(sd, "out.0", h, 1e-5)))
    return modify(h, "after")
