"""TEST INFRASTRUCTURE ONLY.

A deterministic set of transformer_options hooks (every hook point of backend/nn/unet.py:186-279 and :696-763; the two whose arguments are
torch.nn.Module objects -- block_inner_modifiers, group_norm_wrapper -- in build_module_hooks below), written with plain torch ops so the SAME functions run inside the reference UNet on CPU fp32
(fixture generation, oracle/make_golden.py gen_unet_hooks) and inside the native executor on fp16 device tensors (tests/test_gpu_hooks.py).
Each hook also records that it was called, and what `block` / `block_index` / `transformer_index` it saw."""
import torch


def _mha(q, k, v, heads):
    b, nq, c = q.shape
    d = c // heads
    qh, kh, vh = (t.reshape(b, -1, heads, d).transpose(1, 2).float() for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, nq, c).to(q.dtype)


def build_hooks(log=None, use_call_keys=False):
    """-> transformer_options.  `use_call_keys`: the attn2 output patch additionally uses cond_indices / sigmas, which only exist when the
    UNet is driven by sampling_function (sampling_function.py:253-257)."""
    log = log if log is not None else []

    def note(name, to):
        log.append((name, to.get("block"), to.get("block_index"), to.get("transformer_index")))

    def attn1_patch(n, context, value, extra):
        note("attn1_patch", extra)
        return n * 1.05, context * 0.95, value + 0.01 * n

    def attn2_patch(n, context, value, extra):
        note("attn2_patch", extra)
        return n, context * 1.1, value * 0.9

    def attn1_replace(q, k, v, extra):
        note("attn1_replace", extra)
        assert q.shape[-1] == extra["n_heads"] * extra["dim_head"]
        return _mha(q, k, v, extra["n_heads"]) * 0.9

    def attn2_replace(q, k, v, extra):
        note("attn2_replace", extra)
        return _mha(q, k * 1.2, v, extra["n_heads"])

    def attn1_output_patch(n, extra):
        note("attn1_output_patch", extra)
        return n * 0.9

    def attn2_output_patch(n, extra):
        note("attn2_output_patch", extra)
        if use_call_keys:
            n = n.clone()
            n[extra["cond_indices"]] = n[extra["cond_indices"]] * 1.1
            assert extra["sigmas"].shape[0] * len(extra["cond_or_uncond"]) == n.shape[0]
            assert extra["cond_mark"].shape[0] == n.shape[0]
            return n
        return n + 0.02

    def middle_patch(x, extra):
        note("middle_patch", extra)
        return x * 0.98

    def input_block_patch(h, to):
        note("input_block_patch", to)
        return h * 1.02 if to["block"][1] == 1 else h

    def input_block_patch_after_skip(h, to):
        note("input_block_patch_after_skip", to)
        return h * 0.97 if to["block"][1] == 3 else h

    def output_block_patch(h, hsp, to):
        note("output_block_patch", to)
        h2 = h.clone()
        h2[:, :h.shape[1] // 2] = h2[:, :h.shape[1] // 2] * 1.1   # FreeU-style backbone scaling
        return h2, hsp * 0.9

    def block_modifier(h, when, to):
        note("block_modifier_" + when, to)
        blk = to["block"]
        if blk == ("input", 0) and when == "before":
            return h + 0.05
        if blk == ("middle", 0) and when == "after":
            h *= 1.01  # in place
            return h
        if blk == ("output", 1) and when == "before":
            assert h.shape[1] > 0 and to["original_shape"][1] == 4
            return h * 0.99  # sees the concatenated [h, skip]
        if blk == ("last", 0) and when == "after":
            return h * 0.5
        return h
    return {
        "patches": {"attn1_patch": [attn1_patch], "attn2_patch": [attn2_patch], "attn1_output_patch": [attn1_output_patch],
                    "attn2_output_patch": [attn2_output_patch], "middle_patch": [middle_patch], "input_block_patch": [input_block_patch],
                    "input_block_patch_after_skip": [input_block_patch_after_skip], "output_block_patch": [output_block_patch]},
        "patches_replace": {"attn1": {("middle", 0, 0): attn1_replace}, "attn2": {("input", 3): attn2_replace}},
        "block_modifiers": [block_modifier],
    }, log


def build_module_hooks(log=None):
    """The two hooks that receive MODULES: `block_inner_modifiers` (unet.py:73-91: x, 'before' / 'after', layer, layer_index, block,
    transformer_options) and `group_norm_wrapper` (unet.py:436-474, :755-757: norm, x, transformer_options).  They use what extensions use of
    those objects -- the layer's class name, the block's length, the GroupNorm's parameters and the norm itself as a callable -- so the same
    functions run on the reference's modules and on the native executor's stand-ins."""
    log = log if log is not None else []

    def inner(x, when, layer, layer_index, block, to):
        name = type(layer).__name__
        log.append(("inner_" + when, name, layer_index, len(block), to.get("block")))
        if name == "SpatialTransformer" and when == "after":
            return x * 1.03
        if name == "ResBlock" and when == "before":
            return x + 0.02
        if name == "Upsample" and when == "after":
            return x * 0.97
        if name == "Downsample" and when == "before":
            x *= 1.02   # in place
            return x
        if name == "Conv2d" and when == "after":
            return x * 1.01
        return x

    def group_norm_wrapper(norm, x, to):
        log.append(("group_norm_wrapper", int(norm.num_groups), int(norm.num_channels), int(x.shape[1]), to.get("block")))
        y = norm(x)
        y2 = torch.nn.functional.group_norm(x.float(), norm.num_groups, norm.weight.float(), norm.bias.float(), norm.eps).to(x.dtype)
        return (0.5 * (y + y2)) * 0.98

    return {"block_inner_modifiers": [inner], "group_norm_wrapper": group_norm_wrapper}, log
