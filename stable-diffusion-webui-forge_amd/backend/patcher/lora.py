"""LoRA for the native UNet -- counterpart of backend/patcher/lora.py (`model_lora_keys_unet` :43, `load_lora` :19,
`merge_lora_to_weight` :85, `LoraLoader.refresh` :352) and of the key / patch parsing in
packages_3rdparty/comfyui_lora_collection/lora.py (`load_lora` :32, `model_lora_keys_unet` :286).

The native executor keeps its weights in kernel layouts (permuted conv taps, padded heads, fused Q|K, interleaved GEGLU), so
LoRA is an OFFLINE merge into the LDM-layout state dict followed by a re-layout (`IntegratedUNet2DConditionModel._load`),
exactly the "merge" mode of the reference's LoraLoader.  The merge itself is weight-space GEMM work and runs on the MI355X
through the same C-ABI GEMM as everything else:   W' = W * strength_model + (strength * alpha) * up @ down
= fmx_gemm_conv_f16(A = up [out, rank pad 64], W = down^T [in*kh*kw, rank pad 64], alpha, residual = W): fp16 operands, fp32
accumulation, one rounding -- the arithmetic of the reference with computation_dtype fp32 and fp16 LoRA tensors.

Patch types built: "lora" (regular / diffusers / transformers key styles, optional conv `lora_mid`), "diff", "set".
LoHa / LoKr / GLoRA / DoRA files are recognised and rejected explicitly."""
import torch

from ... import hipops as ops
from ..misc.diffusers_state_dict import unet_to_diffusers

UNSUPPORTED_SUFFIXES = (".hada_w1_a", ".lokr_w1", ".lokr_w1_a", ".a1.weight", ".dora_scale")


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


def load_lora(lora, to_load):
    """-> (patch_dict {model key: (type, tensors)}, remaining {unused lora keys}); comfyui_lora_collection/lora.py:32-213 for
    the patch types listed in the module docstring."""
    patch_dict, loaded = {}, set()
    for x, target in to_load.items():
        for suf in UNSUPPORTED_SUFFIXES:
            if x + suf in lora:
                raise NotImplementedError(f"LoRA entry {x}{suf}: LoHa / LoKr / GLoRA / DoRA patches are not supported by the native merge")
        alpha = None
        if x + ".alpha" in lora:
            alpha = float(lora[x + ".alpha"].item())
            loaded.add(x + ".alpha")
        for up, down, mid in ((".lora_up.weight", ".lora_down.weight", ".lora_mid.weight"), ("_lora.up.weight", "_lora.down.weight", None),
                              (".lora_B.weight", ".lora_A.weight", None), (".lora.up.weight", ".lora.down.weight", None),
                              (".lora_linear_layer.up.weight", ".lora_linear_layer.down.weight", None)):
            if x + up in lora:
                m = None
                if mid is not None and x + mid in lora:
                    m = lora[x + mid]
                    loaded.add(x + mid)
                patch_dict[target] = ("lora", (lora[x + up], lora[x + down], alpha, m, None))
                loaded.update((x + up, x + down))
                break
        if x + ".diff" in lora:
            patch_dict[target] = ("diff", (lora[x + ".diff"],))
            loaded.add(x + ".diff")
        if x + ".diff_b" in lora:
            patch_dict[target[:-len(".weight")] + ".bias"] = ("diff", (lora[x + ".diff_b"],))
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
def merge_lora_to_weight(patches, weight, key="online_lora", computation_dtype=torch.float32, device="cuda"):
    """patches: [(strength_patch, (type, tensors) | (tensor,), strength_model, offset, function)] as ModelPatcher.add_patches
    stores them (backend/patcher/base.py); weight: the LDM-layout parameter.  Returns the merged weight, fp16 on `device`.
    Mirrors backend/patcher/lora.py:85-323 for the supported patch types (offset / function hooks are not used by LoRA files)."""
    w = weight.to(device=device, dtype=torch.float16).contiguous()
    shape = w.shape
    for strength, v, strength_model, offset, function in patches:
        if offset is not None or function is not None:
            raise NotImplementedError("weight offset / function hooks are not supported by the native merge")
        if strength_model != 1.0:
            w = ops.scale_f16(w, strength_model) if hasattr(ops, "scale_f16") else (w.float() * strength_model).half()
        if len(v) == 1:
            ptype, v = "diff", v
        else:
            ptype, v = v[0], v[1]
        if ptype == "diff":
            if strength != 0.0:
                d = v[0].to(device=device, dtype=torch.float32)
                if d.shape != w.shape:
                    raise ValueError(f"{key}: diff shape {tuple(d.shape)} != weight shape {tuple(w.shape)}")
                w = (w.float() + strength * d).half()  # elementwise, load time only
        elif ptype == "set":
            w = v[0].to(device=device, dtype=torch.float16).reshape(shape).contiguous()
        elif ptype == "lora":
            up, down, alpha, mid, dora = v
            if dora is not None:
                raise NotImplementedError("DoRA scale")
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
            out = torch.empty_like(w2)
            ops.conv_gemm(a, b, w2.shape[1], alpha=scale, residual=w2, out=out, ld_out=w2.shape[1])
            w = out.reshape(shape)
        else:
            raise NotImplementedError(f"patch type {ptype}")
    return w


def merge_loras_into_state_dict(unet_sd, unet_config, loras, device="cuda"):
    """unet_sd: LDM-layout UNet state dict (no prefix); loras: [(lora_state_dict, strength)] applied in order.
    -> (merged state dict (fp16 on device for touched tensors, untouched entries passed through), report)"""
    key_map = model_lora_keys_unet(list(unet_sd.keys()), unet_config)
    per_key = {}
    unused = []
    for lora_sd, strength in loras:
        patch_dict, remaining = load_lora(lora_sd, key_map)
        unused.append(sorted(k for k in remaining if not k.startswith(("lora_te", "text_encoder", "lora_prior"))))
        for mk, pv in patch_dict.items():
            k = mk[len("diffusion_model."):]
            if k not in unet_sd:
                continue
            per_key.setdefault(k, []).append((float(strength), pv, 1.0, None, None))
    merged = dict(unet_sd)
    for k, patches in per_key.items():
        merged[k] = merge_lora_to_weight(patches, unet_sd[k], key=k, device=device)
    return merged, {"patched": len(per_key), "unused_keys": unused}
