"""LoRA for the native UNet -- counterpart of backend/patcher/lora.py (`model_lora_keys_unet` :43, `load_lora` :19,
`merge_lora_to_weight` :85, `LoraLoader.refresh` :352) and of the key / patch parsing in
packages_3rdparty/comfyui_lora_collection/lora.py (`load_lora` :32, `model_lora_keys_unet` :286).

The native executor keeps its weights in kernel layouts (permuted conv taps, padded heads, fused Q|K, interleaved GEGLU), so
LoRA is an OFFLINE merge into the LDM-layout state dict followed by a re-layout (`IntegratedUNet2DConditionModel._load`),
exactly the "merge" mode of the reference's LoraLoader.  The merge itself is weight-space GEMM work and runs on the MI355X
through the same C-ABI GEMM as everything else:   W' = W * strength_model + (strength * alpha) * up @ down
= fmx_gemm_conv_f16(A = up [out, rank pad 64], W = down^T [in*kh*kw, rank pad 64], alpha, residual = W): fp16 operands, fp32
accumulation, one rounding -- the arithmetic of the reference with computation_dtype fp32 and fp16 LoRA tensors.

Patch types built: "lora" (regular / diffusers / transformers key styles, optional conv `lora_mid`), "diff" (incl. w_norm / b_norm),
"set", and the LyCORIS family "loha", "lokr", "glora", each optionally with a DoRA `dora_scale` (`weight_decompose`, :35-78).  The
LyCORIS deltas are built from small factor products (Hadamard, Kronecker, Tucker cores) in fp32 on the device at load time -- the same
computation_dtype the reference uses -- and added to the weight once."""
import torch

from ... import hipops as ops
from ..misc.diffusers_state_dict import unet_to_diffusers



def model_lora_keys_unet(unet_keys, unet_config, key_map=None):
    """{lora key prefix: model key} for a UNet with LDM parameter names `unet_keys` (no prefix).  Model keys are returned
    with the reference's 'diffusion_model.' prefix.  Mirrors comfyui_lora_collection/lora.py:286-308."""
    key_map = {} if key_map is None else key_map
    for k in unet_keys:
        mk = "diffusion_model." + k
        if k.endswith(".weight"):
            stem = k[:-len(".weight")]
            key_map["lora_unet_" + stem.replace(".", "_")] = mk
            key_map["diffusion_model." + stem] = mk  # generic format
        else:
            key_map[mk] = mk
    for dk, lk in unet_to_diffusers(unet_config).items():
        if not dk.endswith(".weight"):
            continue
        mk = "diffusion_model." + lk
        stem = dk[:-len(".weight")]
        key_map["lora_unet_" + stem.replace(".", "_")] = mk
        key_map["lycoris_" + stem.replace(".", "_")] = mk
        for prefix in ("", "unet."):
            name = prefix + stem.replace(".to_", ".processor.to_")
            if name.endswith(".to_out.0"):
                name = name[:-2]
            key_map[name] = mk
    return key_map


def model_lora_keys_flux(flux_keys, flux_config, key_map=None):
    """{lora key prefix: target} for a Flux transformer with BFL parameter names `flux_keys` (no prefix).  Targets are model keys with the reference's
    'diffusion_model.' prefix, or (key, (dim, offset, size)) / (key, None, function) for the diffusers-named LoRAs whose tensors address one slice of a
    fused projection or map through `swap_scale_shift` (comfyui_lora_collection/lora.py:286-299 for the native names, :342-347 for the three diffusers
    spellings: 'transformer.' (diffusers / simpletuner), 'lycoris_' (simpletuner LyCORIS), 'lora_transformer_' (OneTrainer))."""
    from ..misc.diffusers_state_dict import flux_to_diffusers
    key_map = {} if key_map is None else key_map
    for k in flux_keys:
        mk = "diffusion_model." + k
        if k.endswith(".weight"):
            stem = k[:-len(".weight")]
            key_map["lora_unet_" + stem.replace(".", "_")] = mk
            key_map["diffusion_model." + stem] = mk
        else:
            key_map[mk] = mk
    for dk, to in flux_to_diffusers(flux_config, output_prefix="diffusion_model.").items():
        if not dk.endswith(".weight"):
            continue
        stem = dk[:-len(".weight")]
        key_map["transformer." + stem] = to
        key_map["lycoris_" + stem.replace(".", "_")] = to
        key_map["lora_transformer_" + stem.replace(".", "_")] = to
    return key_map


def load_lora(lora, to_load):
    """-> (patch_dict {model key: (type, tensors)}, remaining {unused lora keys}); comfyui_lora_collection/lora.py:32-213 for
    the patch types listed in the module docstring."""
    patch_dict, loaded = {}, set()
    for x, target in to_load.items():
        alpha = None
        if x + ".alpha" in lora:
            alpha = float(lora[x + ".alpha"].item())
            loaded.add(x + ".alpha")
        dora_scale = lora.get(x + ".dora_scale")
        if dora_scale is not None:
            loaded.add(x + ".dora_scale")
        # where a bias diff of this entry lands.  A slice target `(key, offset[, fn])` (the fused Flux projections) has none: the reference forms
        # "{}.bias".format(to_load[x][:-len(".weight")]) on the TUPLE (lora.py:199-203) -- a key no parameter has -- so its bias diff patches
        # nothing; here the entry is left in `remaining` (reported as unused) instead of raising on tuple + str
        bias_target = target[:-len(".weight")] + ".bias" if isinstance(target, str) else None

        def take(*names):
            out = []
            for n in names:
                t = lora.get(x + n)
                if t is not None:
                    loaded.add(x + n)
                out.append(t)
            return out
        for up, down, mid in ((".lora_up.weight", ".lora_down.weight", ".lora_mid.weight"), ("_lora.up.weight", "_lora.down.weight", None),
                              (".lora_B.weight", ".lora_A.weight", None), (".lora.up.weight", ".lora.down.weight", None),
                              (".lora_linear_layer.up.weight", ".lora_linear_layer.down.weight", None)):
            if x + up in lora:
                m = None
                if mid is not None and x + mid in lora:
                    m = lora[x + mid]
                    loaded.add(x + mid)
                patch_dict[target] = ("lora", (lora[x + up], lora[x + down], alpha, m, dora_scale))
                loaded.update((x + up, x + down))
                break
        if x + ".hada_w1_a" in lora:  # LoHa (comfyui_lora_collection/lora.py:97-118)
            w1a, w1b, w2a, w2b, t1, t2 = take(".hada_w1_a", ".hada_w1_b", ".hada_w2_a", ".hada_w2_b", ".hada_t1", ".hada_t2")
            patch_dict[target] = ("loha", (w1a, w1b, alpha, w2a, w2b, t1, t2, dora_scale))
        lokr = take(".lokr_w1", ".lokr_w2", ".lokr_w1_a", ".lokr_w1_b", ".lokr_w2_a", ".lokr_w2_b", ".lokr_t2")
        if lokr[0] is not None or lokr[1] is not None or lokr[2] is not None or lokr[4] is not None:  # LoKr (:120-160)
            w1, w2, w1_a, w1_b, w2_a, w2_b, t2 = lokr
            patch_dict[target] = ("lokr", (w1, w2, alpha, w1_a, w1_b, w2_a, w2_b, t2, dora_scale))
        if x + ".a1.weight" in lora:  # GLoRA (:162-171)
            a1, a2, b1, b2 = take(".a1.weight", ".a2.weight", ".b1.weight", ".b2.weight")
            patch_dict[target] = ("glora", (a1, a2, b1, b2, alpha, dora_scale))
        if x + ".w_norm" in lora:  # :173-182
            w_norm, b_norm = take(".w_norm", ".b_norm")
            patch_dict[target] = ("diff", (w_norm,))
            if b_norm is not None and bias_target is not None:
                patch_dict[bias_target] = ("diff", (b_norm,))
        if x + ".diff" in lora:
            patch_dict[target] = ("diff", (lora[x + ".diff"],))
            loaded.add(x + ".diff")
        if x + ".diff_b" in lora and bias_target is not None:
            patch_dict[bias_target] = ("diff", (lora[x + ".diff_b"],))
            loaded.add(x + ".diff_b")
        if x + ".set_weight" in lora:
            patch_dict[target] = ("set", (lora[x + ".set_weight"],))
            loaded.add(x + ".set_weight")
    remaining = {k: v for k, v in lora.items() if k not in loaded}
    return patch_dict, remaining


def _pad_k(t):
    """[rows, k] fp16 device -> zero-padded to a multiple of 64 columns (the GEMM's K tile)"""
    k = t.shape[1]
    kp = -(-k // 64) * 64
    if kp == k:
        return t.contiguous()
    out = t.new_zeros(t.shape[0], kp)
    out[:, :k] = t
    return out


@torch.inference_mode()
def merge_lora_to_weight(patches, weight, key="online_lora", computation_dtype=torch.float32, device="cuda", out_dtype=torch.float16):
    """patches: [(strength_patch, (type, tensors) | (tensor,), strength_model, offset, function)] as ModelPatcher.add_patches
    stores them (backend/patcher/base.py); weight: the LDM-layout parameter.  Returns the merged weight, fp16 on `device`.
    Mirrors backend/patcher/lora.py:85-323 for the supported patch types (offset / function hooks are not used by LoRA files)."""
    # storage types other than fp16 (bfloat16 Flux, fp32): the reference casts the weight to fp32, merges, and casts ONCE to the weight's type
    # (patcher/lora.py:85-92, :322) -- a detour through fp16 would flush what lies below 2^-14, overflow above 65504 and round twice
    wide = out_dtype != torch.float16
    w = weight.to(device=device, dtype=torch.float32 if wide else torch.float16).contiguous()
    shape = w.shape
    # Several patches on one key (stacked LoRAs): the reference casts the weight to fp32 once, applies every patch and rounds once at the
    # end (patcher/lora.py:85-92, :322).  So does this: with more than one patch the running weight stays fp32 between patches (`fin` is the
    # identity) and is rounded after the loop; a single patch keeps the fused fp16-residual GEMM (one rounding either way).
    multi = len(patches) > 1 or wide
    fin = (lambda t: t) if multi else (lambda t: t.half())
    if multi:
        w = w.float()
    own = w.data_ptr() != weight.data_ptr() if weight.is_floating_point() and weight.device == w.device else True
    for strength, v, strength_model, offset, function in patches:
        if offset is not None or function is not None:
            # patcher/lora.py:100-108: the patch addresses ONE SLICE of this parameter (the q / k / v / mlp part of a fused Flux projection, which a
            # diffusers-named LoRA patches separately) and / or its delta goes through a function (swap_scale_shift for norm_out.linear).  The slice is
            # merged as a parameter of its own and written back.
            if not own:
                w, own = w.clone(), True
            sl = w if offset is None else w.narrow(offset[0], offset[1], offset[2])
            if function is None:
                sl.copy_(merge_lora_to_weight([(strength, v, strength_model, None, None)], sl.contiguous(), key=key, device=device,
                                              out_dtype=torch.float32 if wide else torch.float16).to(sl.dtype))
                continue
            ptype_f, vf = ("diff", v) if len(v) == 1 else (v[0], v[1])
            if strength_model != 1.0:
                sl.copy_((sl.float() * strength_model).to(sl.dtype))
            if ptype_f == "diff":
                delta = strength * vf[0].to(device=device, dtype=torch.float32)
            elif ptype_f == "lora" and vf[3] is None and vf[4] is None:
                up, down, alpha = vf[0], vf[1], vf[2]
                delta = (strength * ((alpha / down.shape[0]) if alpha is not None else 1.0)) * torch.mm(
                    up.to(device=device, dtype=torch.float32).flatten(1), down.to(device=device, dtype=torch.float32).flatten(1)).reshape(sl.shape)
            else:
                raise NotImplementedError(f"{key}: a {ptype_f} patch through a weight function (only plain lora / diff patches map through swap_scale_shift)")
            sl.copy_((sl.float() + function(delta)).to(sl.dtype))
            continue
        if strength_model != 1.0:
            w = fin(w.float() * strength_model)
        if len(v) == 1:
            ptype, v = "diff", v
        else:
            ptype, v = v[0], v[1]
        if ptype == "diff":
            if strength != 0.0:
                d = v[0].to(device=device, dtype=torch.float32)
                if d.shape != w.shape:
                    raise ValueError(f"{key}: diff shape {tuple(d.shape)} != weight shape {tuple(w.shape)}")
                w = fin(w.float() + strength * d)  # elementwise, load time only
        elif ptype == "set":
            w = v[0].to(device=device, dtype=torch.float32 if multi else torch.float16).reshape(shape).contiguous()
        elif ptype == "lora":
            up, down, alpha, mid, dora = v
            if dora is not None:  # the decomposition needs the delta itself, not only W + delta: fp32 on the device, as the reference
                dn = down.to(device=device, dtype=torch.float32)
                if mid is not None:
                    m = mid.to(device=device, dtype=torch.float32)
                    dn = torch.mm(dn.transpose(0, 1).flatten(1), m.transpose(0, 1).flatten(1)).reshape(dn.shape[1], dn.shape[0], m.shape[2], m.shape[3]).transpose(0, 1)
                diff = torch.mm(up.to(device=device, dtype=torch.float32).flatten(1), dn.flatten(1)).reshape(shape)
                w = fin(_weight_decompose(dora, w, diff, (alpha / down.shape[0]) if alpha is not None else 1.0, strength, device))
                continue
            rank = down.shape[0]
            scale = strength * ((alpha / rank) if alpha is not None else 1.0)
            up2 = up.to(device=device, dtype=torch.float16).flatten(1)                        # [out, rank]
            if mid is not None:  # LoCon with a Tucker mid tensor (:149-152): down' = mid x down
                m = mid.to(device=device, dtype=torch.float32)
                dn = down.to(device=device, dtype=torch.float32)
                dn = torch.mm(dn.transpose(0, 1).flatten(1), m.transpose(0, 1).flatten(1)).reshape(dn.shape[1], dn.shape[0], m.shape[2], m.shape[3]).transpose(0, 1)
                down2 = dn.flatten(1).half()
            else:
                down2 = down.to(device=device, dtype=torch.float16).flatten(1)                # [rank, in*kh*kw]
            w2 = w.reshape(shape[0], -1)
            if up2.shape[0] != w2.shape[0] or down2.shape[1] != w2.shape[1]:
                raise ValueError(f"{key}: LoRA shapes {tuple(up.shape)} x {tuple(down.shape)} do not match weight {tuple(shape)}")
            a = _pad_k(up2)                                   # GEMM "activations": rows = output channels
            b = _pad_k(down2.t().contiguous())                # GEMM "weights":     rows = input features
            if multi:   # delta alone, fp32 out of the GEMM, added to the fp32 running weight
                delta = torch.empty(w2.shape, dtype=torch.float32, device=device)
                ops.conv_gemm(a, b, w2.shape[1], alpha=scale, out=delta, ld_out=w2.shape[1])
                w = (w2 + delta).reshape(shape)
            else:
                out = torch.empty_like(w2)
                ops.conv_gemm(a, b, w2.shape[1], alpha=scale, residual=w2, out=out, ld_out=w2.shape[1])
                w = out.reshape(shape)
        elif ptype in ("loha", "lokr", "glora"):
            f32 = lambda t: t.to(device=device, dtype=torch.float32)
            if ptype == "loha":  # patcher/lora.py:230-266: (w1a w1b) * (w2a w2b), optionally through Tucker cores t1 / t2
                w1a, w1b, alpha, w2a, w2b, t1, t2, dora = v
                scale = (alpha / w1b.shape[0]) if alpha is not None else 1.0
                if t1 is not None:
                    m1 = torch.einsum("i j k l, j r, i p -> p r k l", f32(t1), f32(w1b), f32(w1a))
                    m2 = torch.einsum("i j k l, j r, i p -> p r k l", f32(t2), f32(w2b), f32(w2a))
                else:
                    m1, m2 = torch.mm(f32(w1a), f32(w1b)), torch.mm(f32(w2a), f32(w2b))
                diff = (m1 * m2).reshape(shape)
            elif ptype == "lokr":  # :180-228: kron(w1, w2), each either given or a low-rank product
                w1, w2, alpha, w1_a, w1_b, w2_a, w2_b, t2, dora = v
                dim = None
                if w1 is None:
                    dim = w1_b.shape[0]
                    w1 = torch.mm(f32(w1_a), f32(w1_b))
                else:
                    w1 = f32(w1)
                if w2 is None:
                    dim = w2_b.shape[0]
                    w2 = torch.mm(f32(w2_a), f32(w2_b)) if t2 is None else torch.einsum("i j k l, j r, i p -> p r k l", f32(t2), f32(w2_b), f32(w2_a))
                else:
                    w2 = f32(w2)
                if w2.dim() == 4:
                    w1 = w1.unsqueeze(2).unsqueeze(2)
                scale = (alpha / dim) if (alpha is not None and dim is not None) else 1.0
                diff = torch.kron(w1.contiguous(), w2.contiguous()).reshape(shape)
            else:  # glora :268-306: W a1 a2 + b1 b2 (new layout) or b2 b1 + W a2 a1 (old LyCORIS layout)
                a1, a2, b1, b2, alpha, dora = v
                old = b2.shape[1] == b1.shape[0] == a1.shape[0] == a2.shape[1]
                if b2.shape[0] == b1.shape[1] == a1.shape[1] == a2.shape[0]:
                    if not (old and a2.shape[0] == shape[0] and shape[0] == shape[1]):
                        old = False
                A1, A2, B1, B2 = (f32(t.flatten(1)) for t in (a1, a2, b1, b2))
                scale = 1.0 if alpha is None else (alpha / a1.shape[0] if old else alpha / a2.shape[0])
                wf = w.float()
                if old:
                    diff = (torch.mm(B2, B1) + torch.mm(torch.mm(wf.flatten(1), A2), A1)).reshape(shape)
                else:
                    if wf.dim() > 2:
                        diff = torch.einsum("o i ..., i j -> o j ...", torch.einsum("o i ..., i j -> o j ...", wf, A1), A2).reshape(shape)
                    else:
                        diff = torch.mm(torch.mm(wf, A1), A2).reshape(shape)
                    diff = diff + torch.mm(B1, B2).reshape(shape)
            if dora is not None:
                w = fin(_weight_decompose(dora, w, diff, scale, strength, device))
            else:
                w = fin(w.float() + (strength * scale) * diff)
        else:
            raise NotImplementedError(f"patch type {ptype}")
    return w if w.dtype == out_dtype else w.to(out_dtype)


def _weight_decompose(dora_scale, weight, lora_diff, alpha, strength, device):
    """DoRA (patcher/lora.py:35-78): re-normalise every output (or input) slice of W + alpha * delta to the learned magnitude `dora_scale`,
    then blend towards it by `strength`.  fp32 on the device, fp32 result."""
    ds = dora_scale.to(device=device, dtype=torch.float32)
    wf = weight.float()
    calc = wf + alpha * lora_diff
    if ds.shape[0] == calc.shape[0]:
        # output-axis decomposition: the reference normalises by the norm of the ORIGINAL weight here (:58-63), not of W + delta
        norm = wf.reshape(wf.shape[0], -1).norm(dim=1, keepdim=True).reshape(wf.shape[0], *[1] * (wf.dim() - 1))
    else:
        norm = calc.transpose(0, 1).reshape(calc.shape[1], -1).norm(dim=1, keepdim=True).reshape(calc.shape[1], *[1] * (calc.dim() - 1)).transpose(0, 1)
    norm = norm + torch.finfo(torch.float32).eps  # eps of the computation dtype the weight was cast to (:74)
    calc = calc * (ds / norm)
    out = calc if strength == 1.0 else wf + strength * (calc - wf)
    return out   # fp32; merge_lora_to_weight rounds once per key


def merge_loras_into_state_dict(unet_sd, unet_config, loras, device="cuda", key_map=None, out_dtype=torch.float16):
    """unet_sd: LDM-layout UNet (or BFL-layout Flux transformer) state dict (no prefix); loras: [(lora_state_dict, strength)] applied in order;
    key_map: model_lora_keys_unet(...) by default, model_lora_keys_flux(...) for Flux (targets may carry an offset / a function, ModelPatcher.add_patches
    base.py:99-113).  -> (merged state dict (on device for touched tensors, untouched entries passed through), report)"""
    if key_map is None:
        key_map = model_lora_keys_unet(list(unet_sd.keys()), unet_config)
    per_key = {}
    unused = []
    for lora_sd, strength in loras:
        patch_dict, remaining = load_lora(lora_sd, key_map)
        unused.append(sorted(k for k in remaining if not k.startswith(("lora_te", "text_encoder", "lora_prior"))))
        for target, pv in patch_dict.items():
            offset = function = None
            mk = target
            if isinstance(target, tuple):
                mk, offset = target[0], target[1]
                function = target[2] if len(target) > 2 else None
            k = mk[len("diffusion_model."):]
            if k not in unet_sd:
                continue
            per_key.setdefault(k, []).append((float(strength), pv, 1.0, offset, function))
    merged = dict(unet_sd)
    for k, patches in per_key.items():
        merged[k] = merge_lora_to_weight(patches, unet_sd[k], key=k, device=device, out_dtype=out_dtype)
    return merged, {"patched": len(per_key), "unused_keys": unused}


def merge_loras_into_flux_state_dict(flux_sd, flux_config, loras, device="cuda", dtype=torch.bfloat16):
    """the same offline merge for a Flux transformer (BFL parameter names): native ('lora_unet_double_blocks_0_img_attn_qkv', 'diffusion_model. ...') and
    diffusers-named ('transformer.transformer_blocks.0.attn.to_q', ...) LoRA files.  fp16 storage merges like the UNet (fp16 operands, fp32 accumulation,
    one rounding); bfloat16 storage keeps the running weight in fp32 and is cast once at the end, as the reference does."""
    return merge_loras_into_state_dict(flux_sd, flux_config, loras, device=device, key_map=model_lora_keys_flux(list(flux_sd.keys()), flux_config), out_dtype=dtype)
