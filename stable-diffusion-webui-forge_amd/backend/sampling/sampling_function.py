"""CFG batching -- mirror of backend/sampling/sampling_function.py (`sampling_function` :325, `sampling_function_inner`
:292, `calc_cond_uncond_batch` :154, `sampling_prepare` :366, `sampling_cleanup` :402) for the txt2img hot path.

The reference decides per step, from a free-memory query, whether cond and uncond fit in one UNet batch (:193-213);
on a 288 GB part they always do, so the batch is always [uncond ; cond] (the order the reference produces, :187-229)
and the whole step -- input scaling, UNet, `x - eps*sigma`, area-weighted average (weights 1 over counts 1+1e-37) and
`uncond + (cond - uncond) * cond_scale` -- runs as: one pack kernel, one UNet graph, one combine kernel.
One cond at strength 1 takes that fused path; AND-composed prompts (several conds with strengths), `model_function_wrapper` and the
sampler_pre_cfg / sampler_cfg / sampler_post_cfg / conditioning_modifiers hooks take the general path below (one stacked model call,
weighted averaging, the reference's CFG formula with edit strength).  Python hooks in model_options['transformer_options'] (`patches`, `patches_replace`, `block_modifiers`) are handed to the UNet
executor, which then runs eagerly.  Entries with `area` / `mask` / `timestep_start` / `timestep_end` (regional and time-ranged conditioning,
:17-73) take `_regional_cond_uncond_batch`: one model call per active entry on its rectangle, per-element weighted average.  Inpainting-model
`c_concat` is carried to the executor, which folds it into the first conv once per job.
"""
import math

import torch

from .condition import Condition, compile_conditions, compile_weighted_conditions

def _single(conds, what):
    c = conds[0]
    mc = c["model_conds"]
    return mc["c_crossattn"].cond, (mc["y"].cond if "y" in mc else None), (mc["guidance"].cond if "guidance" in mc else None)


_REGIONAL_KEYS = ("area", "mask", "timestep_start", "timestep_end")


def _is_regional(conds):
    return any(k in c for c in (conds or []) for k in _REGIONAL_KEYS)


def get_area_and_mult(conds, x_in, sigma0):
    """sampling_function.py:17-73: -> None when the entry's sigma window excludes this step, else (area, mult): the latent rectangle
    (h, w, y, x) the entry is evaluated on and its per-element averaging weight = mask * mask_strength * strength, or -- without a mask --
    strength with an 8-row linear feather on every side of the rectangle that is not an edge of the latent."""
    if "timestep_start" in conds and sigma0 > float(conds["timestep_start"]):
        return None
    if "timestep_end" in conds and sigma0 < float(conds["timestep_end"]):
        return None
    area = tuple(int(v) for v in conds["area"]) if "area" in conds else (x_in.shape[2], x_in.shape[3], 0, 0)
    strength = float(conds.get("strength", 1.0))
    b, c = x_in.shape[0], x_in.shape[1]
    h, w, y0, x0 = area
    if "mask" in conds:
        mask = conds["mask"]
        if mask.shape[1] != x_in.shape[2] or mask.shape[2] != x_in.shape[3]:
            raise ValueError(f"cond mask {tuple(mask.shape)} does not match the latent {tuple(x_in.shape)}")
        mask = mask.to(device=x_in.device, dtype=torch.float32)[:, y0:y0 + h, x0:x0 + w] * float(conds.get("mask_strength", 1.0))
        mult = mask.unsqueeze(1).repeat(b // mask.shape[0], c, 1, 1) * strength
    else:
        mult = torch.full((b, c, h, w), strength, dtype=torch.float32, device=x_in.device)
        rr = 8
        ramp = [(1.0 / rr) * (t + 1) for t in range(rr)]
        if y0 != 0:
            for t in range(rr):
                mult[:, :, t:1 + t, :] *= ramp[t]
        if h + y0 < x_in.shape[2]:
            for t in range(rr):
                mult[:, :, h - 1 - t:h - t, :] *= ramp[t]
        if x0 != 0:
            for t in range(rr):
                mult[:, :, :, t:1 + t] *= ramp[t]
        if w + x0 < x_in.shape[3]:
            for t in range(rr):
                mult[:, :, :, w - 1 - t:w - t] *= ramp[t]
    return area, mult


def _regional_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options):
    """calc_cond_uncond_batch (:154-288) for entries with `area` / `mask` / `timestep_start` / `timestep_end` (ComfyUI-style regional and
    time-ranged conditioning; nothing in Forge's own UI produces them, extensions can).  Every active entry is evaluated on its rectangle of
    the latent -- one model call per entry through the general path above -- and the outputs are averaged per element with the weights of
    get_area_and_mult: out = sum_i o_i * m_i / (1e-37 + sum_i m_i), formed as sum_i o_i * w_i with the normalised maps w_i = m_i / count
    (one `fmx_blend_masked` pass per entry).  The maps depend on the set of active entries only; they are rebuilt each step with a handful
    of latent-sized torch ops (set-up arithmetic, like the reference's), which caching per active set would remove."""
    from ... import hipops as ops
    from ..modules.k_model import host_sigmas
    sigma0 = float(host_sigmas(timestep)[0])
    full = (x_in.shape[2], x_in.shape[3], 0, 0)
    preds = []
    for entries, cu in ((cond, 0), (uncond or [], 1)):
        active = []
        for e in entries:
            am = get_area_and_mult(e, x_in, sigma0)
            if am is not None:
                if am[0] != full and "c_concat" in e["model_conds"]:
                    raise NotImplementedError("area conditioning together with an inpainting model's c_concat")
                active.append((e, am[0], am[1]))
        if not active:
            preds.append(torch.zeros_like(x_in))     # 0 / 1e-37 (:155-159, 284-288)
            continue
        count = torch.full_like(x_in, 1e-37)
        for _, (h, w, y0, x0), mult in active:
            count[:, :, y0:y0 + h, x0:x0 + w] += mult
        acc = None
        zeros = None
        for e, (h, w, y0, x0), mult in active:
            crop = x_in if (h, w, y0, x0) == full else x_in[:, :, y0:y0 + h, x0:x0 + w].contiguous()
            single = dict(e)
            single.pop("strength", None)              # the weight is applied below, per element
            c_out, u_out = _general_cond_uncond_batch(model, [single] if cu == 0 else [], [single] if cu == 1 else None, crop, timestep, model_options)
            o = c_out if cu == 0 else u_out
            wmap = torch.zeros_like(x_in)
            wmap[:, :, y0:y0 + h, x0:x0 + w] = mult
            wmap /= count
            if (h, w, y0, x0) != full:                # place the rectangle's output into the frame (memory movement only)
                placed = torch.zeros_like(x_in)
                placed[:, :, y0:y0 + h, x0:x0 + w] = o
                o = placed
            o = o.contiguous()
            if acc is None:
                zeros = torch.zeros_like(x_in)
                acc = ops.blend_masked(o, wmap, o, zeros)
            else:
                if getattr(model, "_ones_like", None) is None or model._ones_like.shape != x_in.shape or model._ones_like.device != x_in.device:
                    model._ones_like = torch.ones_like(x_in)
                acc = ops.blend_masked(acc, model._ones_like, o, wmap, out=acc)
        preds.append(acc)
    return preds[0], preds[1]


def _check_supported(conds, what):
    for c in conds:
        if "gligen" in c:
            raise NotImplementedError(f"{what}: GLIGEN conditioning is not on the native path")


def _fused_ok(model, cond, uncond, model_options):
    """One cond at strength 1 (+ at most one uncond), nothing wrapping the model call: the fused pack -> UNet graph -> combine path."""
    return (len(cond) == 1 and math.isclose(cond[0].get("strength", 1.0), 1.0) and (uncond is None or len(uncond) == 1)
            and "model_function_wrapper" not in model_options and hasattr(model, "denoise_cfg"))


def _general_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options):
    """calc_cond_uncond_batch (sampling_function.py:154-288) for any number of conds / unconds with strengths (AND-composed prompts) and for
    `model_function_wrapper`: every entry covers the whole batch, all of them go through ONE model call stacked in the reference's order
    ([uncond..., cond_K-1, ..., cond_0], :187-229), outputs are averaged with weights = strengths (the accumulators of :155-159)."""
    from ...k_diffusion.sampling import _sigma_vec  # noqa: F401  (SigmaInfo carrier)
    from ..modules.k_model import SigmaInfo, host_sigmas
    entries = [(c, 0) for c in cond] + [(u, 1) for u in (uncond or [])]
    entries.reverse()
    n = len(entries)
    crossattn = [e["model_conds"]["c_crossattn"] for e, _ in entries]
    ys = [e["model_conds"]["y"].cond for e, _ in entries if "y" in e["model_conds"]]
    # the stacked conditioning is the same tensor as long as its parts are (their identity is stable across steps, see condition._gather /
    # prompt_parser._memoised), so the UNet's context cache and the captured graph of this batch size stay valid
    key = tuple((c.cond.data_ptr(), tuple(c.cond.shape)) for c in crossattn) + tuple((t.data_ptr(), tuple(t.shape)) for t in ys)
    cached = getattr(model, "_general_ctx", None)
    if cached is not None and cached[0] == key:
        ctx, y = cached[1], cached[2]
    else:
        ctx = crossattn[0].concat(crossattn[1:])
        y = torch.cat(ys) if ys else None
        model._general_ctx = (key, ctx, y, crossattn, ys)
    cond_or_uncond = [cu for _, cu in entries]
    b = x_in.shape[0]
    input_x = torch.cat([x_in] * n)
    timestep_ = torch.cat([timestep] * n)
    timestep_.fmx_sigma = SigmaInfo(list(host_sigmas(timestep)) * n)
    to = dict(model_options.get("transformer_options", {}))
    to["cond_or_uncond"] = cond_or_uncond[:]
    to["sigmas"] = timestep
    to["cond_mark"] = torch.tensor([float(cu) for cu in cond_or_uncond for _ in range(b)], dtype=timestep.dtype, device=timestep.device)
    to["cond_indices"] = [i * b + j for i, cu in enumerate(cond_or_uncond) if cu == 0 for j in range(b)]
    to["uncond_indices"] = [i * b + j for i, cu in enumerate(cond_or_uncond) if cu != 0 for j in range(b)]
    c = {"c_crossattn": ctx, "transformer_options": to}
    if y is not None:
        c["y"] = y
    cc = entries[0][0]["model_conds"].get("c_concat")
    if cc is not None:
        c["c_concat"] = cc.cond  # identical for every entry (sampling_function.py:344-350); the executor repeats it over the stacked batch
    control = entries[0][0].get("control")
    if control is not None:
        p = control
        while p is not None:
            p.transformer_options = to
            p = p.previous_controlnet
        c["control"] = control.get_control(input_x, timestep_, dict(c), n)
        c["control_model"] = control
    if "model_function_wrapper" in model_options:
        output = model_options["model_function_wrapper"](model.apply_model, {"input": input_x, "timestep": timestep_, "c": c,
                                                                            "cond_or_uncond": cond_or_uncond})
    else:
        output = model.apply_model(input_x, timestep_, **c)
    chunks = output.chunk(n)
    outs = {0: [], 1: []}
    for (e, cu), o in zip(entries, chunks):
        outs[cu].append((float(e.get("strength", 1.0)), o))

    def average(items):
        if not items:
            return torch.zeros_like(x_in) / 1e-37  # zero accumulator over the 1e-37 count (:155-159, 284-288)
        total = sum(w for w, _ in items) + 1e-37
        return ops_lincomb([o.contiguous() for _, o in items], [w / total for w, _ in items])
    return average(outs[0]), average(outs[1])


def ops_lincomb(srcs, coefs):
    from ... import hipops as ops
    return ops.lincomb(srcs, coefs)


def calc_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options, cond_scale=None):
    """-> (cond_pred, uncond_pred) as the reference (:154); on the fused path the CFG combine happens in the same kernel and the combined
    result is returned as a third value when `cond_scale` is given."""
    _check_supported(cond, "cond")
    if uncond is not None:
        _check_supported(uncond, "uncond")
    if _is_regional(cond) or _is_regional(uncond):
        cond_pred, uncond_pred = _regional_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options)
        return None, cond_pred, uncond_pred
    if cond_scale is not None and _fused_ok(model, cond, uncond, model_options):
        cctx = _single(cond, "cond")
        uctx = _single(uncond, "uncond") if uncond is not None else None
        cc = cond[0]["model_conds"].get("c_concat")
        return model.denoise_cfg(x_in, timestep, uctx, cctx, cond_scale, want_parts=True, transformer_options=model_options.get("transformer_options"),
                                 control_model=cond[0].get("control"), **({"c_concat": cc.cond} if cc is not None else {}))
    cond_pred, uncond_pred = _general_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options)
    return None, cond_pred, uncond_pred


def sampling_function_inner(model, x, timestep, uncond, cond, cond_scale, model_options={}, seed=None, return_full=False):
    """:292-322, including the sampler_pre_cfg / sampler_cfg / sampler_post_cfg function hooks and the edit-strength form of the CFG
    formula for AND-composed prompts."""
    from ... import hipops as ops
    edit_strength = sum((item["strength"] if "strength" in item else 1) for item in cond)
    if math.isclose(cond_scale, 1.0) and not model_options.get("disable_cfg1_optimization", False):
        uncond_ = None  # :295-298
    else:
        uncond_ = uncond
    for fn in model_options.get("sampler_pre_cfg_function", []):
        model, cond, uncond_, x, timestep, model_options = fn(model, cond, uncond_, x, timestep, model_options)
    custom_cfg = "sampler_cfg_function" in model_options
    fused_scale = None if (custom_cfg or not math.isclose(edit_strength, 1.0)) else cond_scale
    cfg_result, cond_pred, uncond_pred = calc_cond_uncond_batch(model, cond, uncond_, x, timestep, model_options, fused_scale)
    if custom_cfg:
        args = {"cond": x - cond_pred, "uncond": x - uncond_pred, "cond_scale": cond_scale, "timestep": timestep, "input": x, "sigma": timestep,
                "cond_denoised": cond_pred, "uncond_denoised": uncond_pred, "model": model, "model_options": model_options}
        cfg_result = x - model_options["sampler_cfg_function"](args)
    elif cfg_result is None:
        k = cond_scale * edit_strength if not math.isclose(edit_strength, 1.0) else cond_scale
        cfg_result = ops.lincomb([uncond_pred.contiguous(), cond_pred.contiguous()], [1.0 - k, k])  # uncond + (cond - uncond) * k
    for fn in model_options.get("sampler_post_cfg_function", []):
        args = {"denoised": cfg_result, "cond": cond, "uncond": uncond, "model": model, "uncond_denoised": uncond_pred, "cond_denoised": cond_pred,
                "sigma": timestep, "model_options": model_options, "input": x}
        cfg_result = fn(args)
    if return_full:
        return cfg_result, cond_pred, uncond_pred
    return cfg_result


def sampling_function(self, denoiser_params, cond_scale, cond_composition):
    """Same signature as the reference (:325): `self` is the CFGDenoiser."""
    unet_patcher = self.inner_model.inner_model.forge_objects.unet
    model = unet_patcher.model
    control = unet_patcher.controlnet_linked_list
    x, timestep = denoiser_params.x, denoiser_params.sigma
    uncond = compile_conditions(denoiser_params.text_uncond)
    cond = compile_weighted_conditions(denoiser_params.text_cond, cond_composition)
    image_cond_in = unet_patcher.extra_concat_condition if unet_patcher.extra_concat_condition is not None else denoiser_params.image_cond
    if isinstance(image_cond_in, torch.Tensor) and image_cond_in.shape[0] == x.shape[0] and image_cond_in.shape[2:] == x.shape[2:]:
        for h in (uncond or []) + cond:  # :342-350: inpainting / edit models get the image conditioning as c_concat on both halves
            h["model_conds"]["c_concat"] = Condition(image_cond_in)
    if control is not None:  # :352-357
        for h in cond:
            h["control"] = control
        if uncond is not None:
            for h in uncond:
                h["control"] = control
    model_options, seed = unet_patcher.model_options, self.p.seeds[0]
    for modifier in model_options.get("conditioning_modifiers", []):  # :359-360
        model, x, timestep, uncond, cond, cond_scale, model_options, seed = modifier(model, x, timestep, uncond, cond, cond_scale, model_options, seed)
    return sampling_function_inner(model, x, timestep, uncond, cond, cond_scale, model_options, seed, return_full=True)


def sampling_prepare(unet, x):
    """:366-399: the VRAM juggling (load_models_gpu) has nothing to do here -- weights are resident; what remains is handing every
    ControlNet the predictor and its sigma range (:393-397)."""
    real_model = unet.model
    for cnet in unet.list_controlnets():
        cnet.pre_run(real_model, lambda p: real_model.predictor.percent_to_sigma(p))


def sampling_cleanup(unet):
    for cnet in unet.list_controlnets():
        cnet.cleanup()
