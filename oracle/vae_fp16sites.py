"""TEST INFRASTRUCTURE ONLY (CPU oracle): the VAE decoder with fp16 rounding at the native executor's storage sites.

The companion of oracle/unet_fp16sites.py for SURVEY row a16 (reference backend/nn/vae.py:248-271 Decoder.forward, :77-137 ResnetBlock / AttnBlock, :35-57
Upsample).  The walk is oracle/vae.py's (pinned to the reference's fixtures, tests/test_oracle_golden.py); `rounding=False` reproduces it bit for bit.  With
rounding on, every tensor the native executor (stable-diffusion-webui-forge_amd/backend/nn/vae.py) STORES is rounded to fp16 where it stores it:

  * parameters: convolution weights and biases, GroupNorm gamma / beta (all resident as fp16); the value bias of the mid-block attention folded into
    proj_out's bias in fp32 from the fp16 tensors and rounded once (`_fold_v_bias`);
  * the latent as packed into the channels-last fp16 buffer; post_quant_conv's output; conv_in's output;
  * ResnetBlock: SiLU(GroupNorm(x)) (what conv1 reads), conv1's output, SiLU(GroupNorm(h)), the 1x1 shortcut's output, and conv2 + bias + shortcut with ONE
    rounding (the residual is added in the GEMM epilogue in fp32);  GroupNorm statistics in fp32 from the stored fp16 values;
  * AttnBlock: GroupNorm(x), q and k (with their biases), V^T WITHOUT its bias, q' = fp16(q c^-1/2 log2 e), fp32 scores, P rounded to fp16 for P V with the
    row sum taken over the unrounded exponentials, the attention output, proj_out + folded bias + x with one rounding (the fused 512-wide kernel; other
    widths -- the tiny test networks -- materialise the fp16 score matrix, see _attn);
  * Upsample: the convolution's output (the nearest resize happens on load, no tensor);  SiLU(GroupNorm(h)) in front of conv_out, conv_out's output.

`teacher` / `layer_out` are the layer-wise comparison's two halves (tests/test_gpu_vae_sharp_parity.py): with `teacher` = the native executor's recorded layer
outputs (NCHW fp32 views), every layer is evaluated on the NATIVE output of the layer in front of it, so that a rounding flip in layer 3 is not an input
difference for layer 30 (why a whole-network comparison cannot be sharp: DESIGN.md 2.4), and `layer_out` receives what the oracle computes for each layer.
Keys: "conv_in", "<resblock>.h" (conv1's output), "<resblock>", "<attn>.q / .k / .v (V^T without bias, as [B, C, N]) / .o", "<attn>",
"<upsample key>", "conv_out".  `plant`: {"gn_eps": (resblock key, "norm1" | "norm2", eps)} evaluates ONE GroupNorm with another epsilon (a planted bug).
parity: pinned through oracle/vae.py (rounding off is that walk) and against the reference's own fp16 floor with rounding on (tests/test_oracle_vae_fp16sites.py).
"""
import torch
import torch.nn.functional as F

LOG2E = 1.4426950408889634


class _State:
    def __init__(self, rounding, teacher, layer_out, plant, up2x=()):
        self.rounding, self.teacher, self.layer_out, self.plant = rounding, teacher, layer_out, plant or {}
        self.up2x = set(up2x or ())
        self.R = (lambda t: t.half().float()) if rounding else (lambda t: t)

    def teach(self, key, computed):
        """record what the oracle computed for layer `key`; hand the NEXT layer the teacher's version of it when there is one"""
        if self.layer_out is not None:
            self.layer_out[key] = computed
        if self.teacher is not None and key in self.teacher:
            return self.teacher[key].float()
        return computed


def _gn(st, sd, key, x, eps=1e-6):
    """GroupNorm32 (vae.py:12-13) with fp16 gamma / beta; statistics in fp32 from the values as stored"""
    return F.group_norm(x, 32, st.R(sd[key + ".weight"]), st.R(sd[key + ".bias"]), eps)


def _conv(st, sd, key, x, padding=1, bias=True):
    return F.conv2d(x, st.R(sd[key + ".weight"]), st.R(sd[key + ".bias"]) if bias else None, padding=padding)


def _resnet(st, sd, key, x):
    R = st.R
    p = st.plant.get("gn_eps")
    eps1 = p[2] if p and p[0] == key and p[1] == "norm1" else 1e-6
    eps2 = p[2] if p and p[0] == key and p[1] == "norm2" else 1e-6
    g1 = R(F.silu(_gn(st, sd, key + ".norm1", x, eps1)))
    h = st.teach(key + ".h", R(_conv(st, sd, key + ".conv1", g1)))
    g2 = R(F.silu(_gn(st, sd, key + ".norm2", h, eps2)))
    h2 = _conv(st, sd, key + ".conv2", g2)
    if key + ".nin_shortcut.weight" in sd:
        x = R(_conv(st, sd, key + ".nin_shortcut", x, 0))
    return st.teach(key, R(x + h2))


def _attn(st, sd, key, x):
    """AttnBlock (vae.py:99-137): one head as wide as the channel count"""
    from .attention import attention_single_head_spatial
    R = st.R
    b, c, hh, ww = x.shape
    g = R(_gn(st, sd, key + ".norm", x))
    q = st.teach(key + ".q", R(_conv(st, sd, key + ".q", g, 0)))
    k = st.teach(key + ".k", R(_conv(st, sd, key + ".k", g, 0)))
    if not st.rounding:      # oracle/vae.py's own arithmetic, bit for bit
        v = _conv(st, sd, key + ".v", g, 0)
        o = st.teach(key + ".o", attention_single_head_spatial(q, k, v))
        return st.teach(key, x + _conv(st, sd, key + ".proj_out", o, 0))
    v = st.teach(key + ".v", R(_conv(st, sd, key + ".v", g, 0, bias=False)).reshape(b, c, hh * ww))      # V^T without its bias, [B, C, N]
    qt = q.reshape(b, c, hh * ww).transpose(1, 2)                                                        # [B, N, C]
    kt = k.reshape(b, c, hh * ww)                                                                        # [B, C, N]
    if c != 512:
        # widths other than the fused kernel's 512 (the tiny test networks): the executor materialises S = fp16(scale Q K^T), softmaxes its rows in place
        # (fp32 arithmetic on the fp16 scores, fp16 result), and runs P V + b_v and proj_out (+ its own bias, + x) as two GEMMs
        s = R(torch.matmul(qt, kt) * (c ** -0.5))
        p = R(s.softmax(dim=-1))
        o = R(torch.matmul(p, v.transpose(1, 2)) + R(sd[key + ".v.bias"]))
        o = st.teach(key + ".o", o.transpose(1, 2).reshape(b, c, hh, ww))
        return st.teach(key, R(_conv(st, sd, key + ".proj_out", o, 0) + x))
    qs = R(qt * (torch.tensor(c ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)))
    o = torch.empty_like(qt)
    step = max(1, (1 << 26) // (hh * ww))
    for i in range(0, hh * ww, step):
        s = torch.matmul(qs[:, i:i + step], kt)
        p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
        o[:, i:i + step] = torch.matmul(R(p), v.transpose(1, 2)) / p.sum(dim=-1, keepdim=True)
    o = st.teach(key + ".o", R(o).transpose(1, 2).reshape(b, c, hh, ww))
    w_o = R(sd[key + ".proj_out.weight"]).reshape(c, c)
    folded = R(R(sd[key + ".proj_out.bias"]) + w_o @ R(sd[key + ".v.bias"]))                            # _fold_v_bias: fp32 from the fp16 tensors, one rounding
    return st.teach(key, R(F.conv2d(o, R(sd[key + ".proj_out.weight"]), folded) + x))


@torch.no_grad()
def vae_decode(sd, z, rounding=True, teacher=None, layer_out=None, plant=None, up2x=()):
    """IntegratedAutoencoderKL.decode on z [B, lc, h, w] (already process_out'ed) -> [B, 3, 8h, 8w] (vae.py:305-316 without the scaling)."""
    st = _State(rounding, teacher, layer_out, plant, up2x)
    R = st.R
    h = R(z.float())
    if "post_quant_conv.weight" in sd:
        h = R(_conv(st, sd, "post_quant_conv", h, 0))
    h = st.teach("conv_in", R(_conv(st, sd, "decoder.conv_in", h)))
    h = _resnet(st, sd, "decoder.mid.block_1", h)
    h = _attn(st, sd, "decoder.mid.attn_1", h)
    h = _resnet(st, sd, "decoder.mid.block_2", h)
    nlev = 0
    while f"decoder.up.{nlev}.block.0.norm1.weight" in sd:
        nlev += 1
    for lev in reversed(range(nlev)):
        i = 0
        while f"decoder.up.{lev}.block.{i}.norm1.weight" in sd:
            h = _resnet(st, sd, f"decoder.up.{lev}.block.{i}", h)
            i += 1
        if lev != 0:
            up = f"decoder.up.{lev}.upsample"
            hu = F.interpolate(h, scale_factor=2.0, mode="nearest")
            if rounding and up in st.up2x:
                # the executor ran this Upsample convolution as four phase convolutions on tap-summed weights (IntegratedAutoencoderKL.up2x_trace;
                # oracle/unet_fp16sites.py up2x_phase_conv): the sums of the rounded taps are rounded once more
                from .unet_fp16sites import up2x_phase_conv
                h = st.teach(up, R(up2x_phase_conv(hu, R(sd[up + ".conv.weight"]), R(sd[up + ".conv.bias"]), R)))
            else:
                h = st.teach(up, R(_conv(st, sd, up + ".conv", hu)))
    g = R(F.silu(_gn(st, sd, "decoder.norm_out", h)))
    return st.teach("conv_out", R(_conv(st, sd, "decoder.conv_out", g)))


def kind_of(key):
    """coarse layer class of a tap key, for the per-kind summary of the comparison"""
    if key.endswith(".h"):
        return "ResnetBlock conv1"
    if key.endswith((".q", ".k", ".v")):
        return "attention q / k / V^T"
    if key.endswith(".o"):
        return "attention output"
    if key.endswith("attn_1"):
        return "AttnBlock output"
    if key.endswith(".upsample"):
        return "Upsample convolution"
    if key in ("conv_in", "conv_out"):
        return key
    return "ResnetBlock output"
