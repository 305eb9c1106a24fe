"""Static structure of the LDM UNet / AutoencoderKL decoder, derived from the same config dicts the
reference's constructors take (backend/nn/unet.py:485-694, backend/nn/vae.py:203-246,277-294).

The reference builds nn.Modules; the MI355X executor instead walks a flat, immutable layout (this
file) and binds checkpoint tensors by their LDM key names, so the same layout also enumerates the
expected state-dict keys and shapes (`unet_param_shapes`, `vae_decoder_param_shapes`).
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass(frozen=True)
class ConvIn:
    key: str
    cin: int
    cout: int


@dataclass(frozen=True)
class Res:
    key: str
    cin: int
    cout: int

    @property
    def has_skip_conv(self):
        return self.cin != self.cout


@dataclass(frozen=True)
class SpatialT:
    key: str
    ch: int
    heads: int
    dim_head: int
    depth: int
    context_dim: int
    use_linear: bool


@dataclass(frozen=True)
class Down:
    key: str
    ch: int


@dataclass(frozen=True)
class Up:
    key: str
    ch: int


@dataclass
class UNetLayout:
    in_channels: int
    model_channels: int
    out_channels: int
    time_embed_dim: int
    adm_in_channels: Optional[int]
    context_dim: int
    input_blocks: List[list] = field(default_factory=list)
    middle: list = field(default_factory=list)
    output_blocks: List[list] = field(default_factory=list)
    out_ch: int = 0

    def all_layers(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle
        for blk in self.output_blocks:
            yield from blk


def _heads(ch, num_heads, num_head_channels):
    # unet.py:562-566
    if num_head_channels == -1:
        return num_heads, ch // num_heads
    return ch // num_head_channels, num_head_channels


def unet_layout(cfg, encoder_only=False) -> UNetLayout:
    """Enumerate blocks exactly as IntegratedUNet2DConditionModel.__init__ does (unet.py:540-694).  encoder_only: input blocks +
    middle block, the trunk of cldm.ControlNet (backend/nn/cnets/cldm.py:101-204: same enumeration, no decoder)."""
    mc = cfg["model_channels"]
    channel_mult = tuple(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(channel_mult) if isinstance(nrb, int) else list(nrb)
    td = list(cfg["transformer_depth"])
    td_out = list(cfg.get("transformer_depth_output") or []) if encoder_only else list(cfg["transformer_depth_output"])
    td_mid = cfg["transformer_depth_middle"]
    num_heads = cfg.get("num_heads", -1)
    nhc = cfg.get("num_head_channels", -1)
    ctx = cfg["context_dim"]
    use_linear = bool(cfg.get("use_linear_in_transformer", False))
    if cfg.get("use_scale_shift_norm") or cfg.get("resblock_updown") or cfg.get("num_attention_blocks") \
            or cfg.get("disable_self_attentions") or cfg.get("dims", 2) != 2:
        raise NotImplementedError("UNet option outside the SD1.x/SD2.x/SDXL family")
    num_classes = cfg.get("num_classes")
    if num_classes not in (None, "sequential"):
        raise NotImplementedError("only num_classes None / 'sequential' (unet.py:524-538)")

    lay = UNetLayout(in_channels=cfg["in_channels"], model_channels=mc, out_channels=cfg.get("out_channels", cfg["in_channels"]),
                     time_embed_dim=mc * 4,
                     adm_in_channels=cfg.get("adm_in_channels") if num_classes == "sequential" else None,
                     context_dim=ctx)
    lay.input_blocks.append([ConvIn("input_blocks.0.0", cfg["in_channels"], mc)])
    chans = [mc]
    ch = mc
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(nrb[level]):
            blk = [Res(f"input_blocks.{idx}.0", ch, mult * mc)]
            ch = mult * mc
            depth = td.pop(0)
            if depth > 0:
                h, d = _heads(ch, num_heads, nhc)
                blk.append(SpatialT(f"input_blocks.{idx}.1", ch, h, d, depth, ctx, use_linear))
            lay.input_blocks.append(blk)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            lay.input_blocks.append([Down(f"input_blocks.{idx}.0", ch)])
            chans.append(ch)
            idx += 1
    h, d = _heads(ch, num_heads, nhc)
    lay.middle = [Res("middle_block.0", ch, ch)]
    if td_mid >= 0:
        lay.middle += [SpatialT("middle_block.1", ch, h, d, td_mid, ctx, use_linear),
                       Res("middle_block.2", ch, ch)]
    lay.out_ch = ch
    if encoder_only:
        lay.zero_conv_channels = chans  # one zero conv per input block (cldm.py:108,166,190), + middle_block_out on `ch`
        return lay
    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(nrb[level] + 1):
            ich = chans.pop()
            blk = [Res(f"output_blocks.{idx}.0", ch + ich, mc * mult)]
            ch = mc * mult
            depth = td_out.pop()
            sub = 1
            if depth > 0:
                h, d = _heads(ch, num_heads, nhc)
                blk.append(SpatialT(f"output_blocks.{idx}.{sub}", ch, h, d, depth, ctx, use_linear))
                sub += 1
            if level and i == nrb[level]:
                blk.append(Up(f"output_blocks.{idx}.{sub}", ch))
            lay.output_blocks.append(blk)
            idx += 1
    lay.out_ch = ch
    return lay


HINT_BLOCK = ((16, 1), (16, 1), (32, 2), (32, 1), (96, 2), (96, 1), (256, 2))  # (out channels, stride) of cldm.py:109-125, then -> model_channels


def controlnet_param_shapes(cfg, hint_channels=3):
    """cldm.ControlNet (backend/nn/cnets/cldm.py:74-207): the UNet encoder's keys + input_hint_block + zero_convs + middle_block_out."""
    s = unet_param_shapes(cfg, encoder_only=True)
    lay = unet_layout(cfg, encoder_only=True)
    cin = hint_channels
    for i, (cout, _) in enumerate(HINT_BLOCK + ((lay.model_channels, 1),)):
        s[f"input_hint_block.{2 * i}.weight"], s[f"input_hint_block.{2 * i}.bias"] = (cout, cin, 3, 3), (cout,)
        cin = cout
    for i, c in enumerate(lay.zero_conv_channels):
        s[f"zero_convs.{i}.0.weight"], s[f"zero_convs.{i}.0.bias"] = (c, c, 1, 1), (c,)
    s["middle_block_out.0.weight"], s["middle_block_out.0.bias"] = (lay.out_ch, lay.out_ch, 1, 1), (lay.out_ch,)
    return s


def unet_param_shapes(cfg, encoder_only=False) -> "OrderedDict[str, Tuple[int, ...]]":
    lay = unet_layout(cfg, encoder_only)
    s = OrderedDict()
    te = lay.time_embed_dim

    def lin(key, i, o, bias=True):
        s[key + ".weight"] = (o, i)
        if bias:
            s[key + ".bias"] = (o,)

    def conv(key, i, o, k):
        s[key + ".weight"] = (o, i, k, k)
        s[key + ".bias"] = (o,)

    def norm(key, c):
        s[key + ".weight"] = (c,)
        s[key + ".bias"] = (c,)

    lin("time_embed.0", lay.model_channels, te)
    lin("time_embed.2", te, te)
    if lay.adm_in_channels is not None:
        lin("label_emb.0.0", lay.adm_in_channels, te)
        lin("label_emb.0.2", te, te)
    for L in lay.all_layers():
        if isinstance(L, ConvIn):
            conv(L.key, L.cin, L.cout, 3)
        elif isinstance(L, Res):
            norm(L.key + ".in_layers.0", L.cin)
            conv(L.key + ".in_layers.2", L.cin, L.cout, 3)
            lin(L.key + ".emb_layers.1", te, L.cout)
            norm(L.key + ".out_layers.0", L.cout)
            conv(L.key + ".out_layers.3", L.cout, L.cout, 3)
            if L.has_skip_conv:
                conv(L.key + ".skip_connection", L.cin, L.cout, 1)
        elif isinstance(L, SpatialT):
            inner = L.heads * L.dim_head
            norm(L.key + ".norm", L.ch)
            if L.use_linear:
                lin(L.key + ".proj_in", L.ch, inner)
            else:
                conv(L.key + ".proj_in", L.ch, inner, 1)
            for d in range(L.depth):
                b = f"{L.key}.transformer_blocks.{d}"
                for a, cdim in (("attn1", inner), ("attn2", L.context_dim)):
                    lin(f"{b}.{a}.to_q", inner, inner, bias=False)
                    lin(f"{b}.{a}.to_k", cdim, inner, bias=False)
                    lin(f"{b}.{a}.to_v", cdim, inner, bias=False)
                    lin(f"{b}.{a}.to_out.0", inner, inner)
                lin(f"{b}.ff.net.0.proj", inner, inner * 8)
                lin(f"{b}.ff.net.2", inner * 4, inner)
                for n in ("norm1", "norm2", "norm3"):
                    norm(f"{b}.{n}", inner)
            if L.use_linear:
                lin(L.key + ".proj_out", inner, L.ch)
            else:
                conv(L.key + ".proj_out", inner, L.ch, 1)
        elif isinstance(L, Down):
            conv(L.key + ".op", L.ch, L.ch, 3)
        elif isinstance(L, Up):
            conv(L.key + ".conv", L.ch, L.ch, 3)
    if encoder_only:
        return s
    norm("out.0", lay.out_ch)
    conv("out.2", lay.model_channels, lay.out_channels, 3)
    return s


# ------------------------------------------------------------------------------------------------
# AutoencoderKL decoder (vae.py:203-271, 277-294)
# ------------------------------------------------------------------------------------------------

@dataclass
class VaeDecoderLayout:
    latent_channels: int
    out_channels: int
    block_in: int
    use_post_quant_conv: bool
    scaling_factor: float
    shift_factor: float
    # list over execution order (i_level reversed): (i_level, [(key, cin, cout)...], upsample_key or None)
    levels: list = field(default_factory=list)
    final_ch: int = 0


def vae_decoder_layout(cfg) -> VaeDecoderLayout:
    boc = list(cfg["block_out_channels"])
    ch = boc[0]
    ch_mult = [c // ch for c in boc]
    nres = cfg["layers_per_block"]
    nlev = len(ch_mult)
    block_in = ch * ch_mult[-1]
    shift = cfg.get("shift_factor", 0.0)
    lay = VaeDecoderLayout(latent_channels=cfg.get("latent_channels", 4), out_channels=cfg.get("out_channels", 3),
                           block_in=block_in, use_post_quant_conv=cfg.get("use_post_quant_conv", True),
                           scaling_factor=cfg.get("scaling_factor", 0.18215),
                           shift_factor=shift if isinstance(shift, float) else 0.0)
    cin = block_in
    for i_level in reversed(range(nlev)):
        cout = ch * ch_mult[i_level]
        blocks = []
        for i_block in range(nres + 1):
            blocks.append((f"decoder.up.{i_level}.block.{i_block}", cin, cout))
            cin = cout
        up = f"decoder.up.{i_level}.upsample" if i_level != 0 else None
        lay.levels.append((i_level, blocks, up))
    lay.final_ch = cin
    return lay


def vae_decoder_param_shapes(cfg):
    lay = vae_decoder_layout(cfg)
    s = OrderedDict()

    def conv(key, i, o, k):
        s[key + ".weight"] = (o, i, k, k)
        s[key + ".bias"] = (o,)

    def norm(key, c):
        s[key + ".weight"] = (c,)
        s[key + ".bias"] = (c,)

    def res(key, i, o):
        norm(key + ".norm1", i)
        conv(key + ".conv1", i, o, 3)
        norm(key + ".norm2", o)
        conv(key + ".conv2", o, o, 3)
        if i != o:
            conv(key + ".nin_shortcut", i, o, 1)

    if lay.use_post_quant_conv:
        conv("post_quant_conv", lay.latent_channels, lay.latent_channels, 1)
    bi = lay.block_in
    conv("decoder.conv_in", lay.latent_channels, bi, 3)
    res("decoder.mid.block_1", bi, bi)
    norm("decoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"decoder.mid.attn_1.{n}", bi, bi, 1)
    res("decoder.mid.block_2", bi, bi)
    for _, blocks, up in lay.levels:
        for key, i, o in blocks:
            res(key, i, o)
        if up is not None:
            c = blocks[-1][2]
            conv(up + ".conv", c, c, 3)
    norm("decoder.norm_out", lay.final_ch)
    conv("decoder.conv_out", lay.final_ch, lay.out_channels, 3)
    return s


# ------------------------------------------------------------------------------------------------
# AutoencoderKL encoder (vae.py:140-200, Downsample :60-74, quant_conv :287, encode :296-303) -- img2img / hires path
# ------------------------------------------------------------------------------------------------

@dataclass
class VaeEncoderLayout:
    in_channels: int
    latent_channels: int
    ch: int
    block_in: int
    use_quant_conv: bool
    # list over execution order: (i_level, [(key, cin, cout)...], downsample_key or None)
    levels: list = field(default_factory=list)


def vae_encoder_layout(cfg) -> VaeEncoderLayout:
    boc = list(cfg["block_out_channels"])
    ch = boc[0]
    ch_mult = [c // ch for c in boc]
    nres = cfg["layers_per_block"]
    nlev = len(ch_mult)
    lay = VaeEncoderLayout(in_channels=cfg.get("in_channels", 3), latent_channels=cfg.get("latent_channels", 4), ch=ch,
                           block_in=ch * ch_mult[-1], use_quant_conv=cfg.get("use_quant_conv", True))
    in_ch_mult = [1] + ch_mult
    for i_level in range(nlev):
        cin = ch * in_ch_mult[i_level]
        cout = ch * ch_mult[i_level]
        blocks = []
        for i_block in range(nres):
            blocks.append((f"encoder.down.{i_level}.block.{i_block}", cin, cout))
            cin = cout
        down = f"encoder.down.{i_level}.downsample" if i_level != nlev - 1 else None
        lay.levels.append((i_level, blocks, down))
    return lay


def vae_encoder_param_shapes(cfg):
    lay = vae_encoder_layout(cfg)
    s = OrderedDict()

    def conv(key, i, o, k):
        s[key + ".weight"] = (o, i, k, k)
        s[key + ".bias"] = (o,)

    def norm(key, c):
        s[key + ".weight"] = (c,)
        s[key + ".bias"] = (c,)

    def res(key, i, o):
        norm(key + ".norm1", i)
        conv(key + ".conv1", i, o, 3)
        norm(key + ".norm2", o)
        conv(key + ".conv2", o, o, 3)
        if i != o:
            conv(key + ".nin_shortcut", i, o, 1)

    conv("encoder.conv_in", lay.in_channels, lay.ch, 3)
    for _, blocks, down in lay.levels:
        for key, i, o in blocks:
            res(key, i, o)
        if down is not None:
            c = blocks[-1][2]
            conv(down + ".conv", c, c, 3)
    bi = lay.block_in
    res("encoder.mid.block_1", bi, bi)
    norm("encoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"encoder.mid.attn_1.{n}", bi, bi, 1)
    res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi)
    conv("encoder.conv_out", bi, 2 * lay.latent_channels, 3)
    if lay.use_quant_conv:
        conv("quant_conv", 2 * lay.latent_channels, 2 * lay.latent_channels, 1)
    return s


# ------------------------------------------------------------------------------------------------
# CLIP text encoder (backend/nn/clip.py IntegratedCLIP over transformers' CLIPTextModel; key names as in Forge checkpoints)
# ------------------------------------------------------------------------------------------------

def clip_param_shapes(cfg):
    """cfg: hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, vocab_size, max_position_embeddings,
    hidden_act ('quick_gelu' | 'gelu'), optional add_text_projection"""
    c, f = cfg["hidden_size"], cfg["intermediate_size"]
    s = OrderedDict()
    p = "transformer.text_model."
    s[p + "embeddings.token_embedding.weight"] = (cfg["vocab_size"], c)
    s[p + "embeddings.position_embedding.weight"] = (cfg["max_position_embeddings"], c)
    for i in range(cfg["num_hidden_layers"]):
        b = f"{p}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[b + f"self_attn.{n}.weight"] = (c, c)
            s[b + f"self_attn.{n}.bias"] = (c,)
        for n in ("layer_norm1", "layer_norm2"):
            s[b + n + ".weight"] = (c,)
            s[b + n + ".bias"] = (c,)
        s[b + "mlp.fc1.weight"] = (f, c)
        s[b + "mlp.fc1.bias"] = (f,)
        s[b + "mlp.fc2.weight"] = (c, f)
        s[b + "mlp.fc2.bias"] = (c,)
    s[p + "final_layer_norm.weight"] = (c,)
    s[p + "final_layer_norm.bias"] = (c,)
    if cfg.get("add_text_projection"):
        s["transformer.text_projection.weight"] = (c, c)
    return s


def t5_param_shapes(cfg):
    """State-dict keys / shapes of IntegratedT5 (backend/nn/t5.py:196-214) for a config with the reference's keys: d_model, d_ff, num_layers, num_heads,
    vocab_size, dense_act_fn, is_gated_act, model_type.  The reference builds the attention at inner_dim = d_model (:186); the relative-position bias
    table (32 buckets x heads) sits in block 0 only unless model_type == 'umt5' (:179-181)."""
    c, f, H = cfg["d_model"], cfg["d_ff"], cfg["num_heads"]
    s = OrderedDict()
    s["transformer.shared.weight"] = (cfg["vocab_size"], c)
    for i in range(cfg["num_layers"]):
        b = f"transformer.encoder.block.{i}.layer."
        for n in ("q", "k", "v", "o"):
            s[b + f"0.SelfAttention.{n}.weight"] = (c, c)
        if i == 0 or cfg.get("model_type") == "umt5":
            s[b + "0.SelfAttention.relative_attention_bias.weight"] = (32, H)
        s[b + "0.layer_norm.weight"] = (c,)
        if cfg.get("is_gated_act", True):
            s[b + "1.DenseReluDense.wi_0.weight"] = (f, c)
            s[b + "1.DenseReluDense.wi_1.weight"] = (f, c)
        else:
            s[b + "1.DenseReluDense.wi.weight"] = (f, c)
        s[b + "1.DenseReluDense.wo.weight"] = (c, f)
        s[b + "1.layer_norm.weight"] = (c,)
    s["transformer.encoder.final_layer_norm.weight"] = (c,)
    s["logit_scale"] = ()
    return s


# ---- Flux (MMDiT) ---------------------------------------------------------------------------------------------------
def flux_param_shapes(cfg):
    """State-dict keys / shapes of IntegratedFluxTransformer2DModel (backend/nn/flux.py:310-367) for a config dict with the
    reference ctor's keys (in_channels, vec_in_dim, context_in_dim, hidden_size, mlp_ratio, num_heads, depth,
    depth_single_blocks, axes_dim, theta, qkv_bias, guidance_embed)."""
    hs, nh = cfg["hidden_size"], cfg["num_heads"]
    hd = hs // nh
    mlp = int(hs * cfg["mlp_ratio"])
    inc = cfg["in_channels"] * 4
    out = OrderedDict()

    def lin(k, nin, nout, bias=True):
        out[k + ".weight"] = (nout, nin)
        if bias:
            out[k + ".bias"] = (nout,)

    lin("img_in", inc, hs)
    lin("time_in.in_layer", 256, hs)
    lin("time_in.out_layer", hs, hs)
    lin("vector_in.in_layer", cfg["vec_in_dim"], hs)
    lin("vector_in.out_layer", hs, hs)
    if cfg["guidance_embed"]:
        lin("guidance_in.in_layer", 256, hs)
        lin("guidance_in.out_layer", hs, hs)
    lin("txt_in", cfg["context_in_dim"], hs)
    for i in range(cfg["depth"]):
        b = f"double_blocks.{i}"
        for st in ("img", "txt"):
            lin(f"{b}.{st}_mod.lin", hs, 6 * hs)
            lin(f"{b}.{st}_attn.qkv", hs, 3 * hs, bias=cfg["qkv_bias"])
            out[f"{b}.{st}_attn.norm.query_norm.scale"] = (hd,)
            out[f"{b}.{st}_attn.norm.key_norm.scale"] = (hd,)
            lin(f"{b}.{st}_attn.proj", hs, hs)
            lin(f"{b}.{st}_mlp.0", hs, mlp)
            lin(f"{b}.{st}_mlp.2", mlp, hs)
    for i in range(cfg["depth_single_blocks"]):
        b = f"single_blocks.{i}"
        lin(f"{b}.linear1", hs, 3 * hs + mlp)
        lin(f"{b}.linear2", hs + mlp, hs)
        out[f"{b}.norm.query_norm.scale"] = (hd,)
        out[f"{b}.norm.key_norm.scale"] = (hd,)
        lin(f"{b}.modulation.lin", hs, 3 * hs)
    lin("final_layer.linear", hs, inc)
    lin("final_layer.adaLN_modulation.1", hs, 2 * hs)
    return out
