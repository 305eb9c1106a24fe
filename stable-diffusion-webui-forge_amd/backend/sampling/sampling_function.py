"""CFG batching -- mirror of backend/sampling/sampling_function.py (`sampling_function` :325, `sampling_function_inner`
:292, `calc_cond_uncond_batch` :154, `sampling_prepare` :366, `sampling_cleanup` :402) for the txt2img hot path.

The reference decides per step, from a free-memory query, whether cond and uncond fit in one UNet batch (:193-213);
on a 288 GB part they always do, so the batch is always [uncond ; cond] (the order the reference produces, :187-229)
and the whole step -- input scaling, UNet, `x - eps*sigma`, area-weighted average (weights 1 over counts 1+1e-37) and
`uncond + (cond - uncond) * cond_scale` -- runs as: one pack kernel, one UNet graph, one combine kernel.
Python hooks in model_options['transformer_options'] (`patches`, `patches_replace`, `block_modifiers`) are handed to the UNet
executor, which then runs eagerly.  Features that would need several UNet calls per step (regional `area` conds, masks, per-cond
timestep ranges, c_concat) are rejected explicitly.
"""
import math

import torch

from .condition import compile_conditions, compile_weighted_conditions

_UNSUPPORTED_OPTS = ("model_function_wrapper", "sampler_cfg_function", "sampler_pre_cfg_function",
                     "sampler_post_cfg_function", "conditioning_modifiers")


def _single(conds, what):
    if len(conds) != 1:
        raise NotImplementedError(f"{what}: composable / AND prompts need several UNet passes per step; not on the native path")
    c = conds[0]
    for k in ("area", "mask", "timestep_start", "timestep_end"):
        if k in c:
            raise NotImplementedError(f"{what}: '{k}' conditioning is not supported by the native path")
    if not math.isclose(c.get("strength", 1.0), 1.0):
        raise NotImplementedError("prompt weights != 1 change the CFG formula (edit_strength); not on the native path")
    mc = c["model_conds"]
    return mc["c_crossattn"].cond, (mc["y"].cond if "y" in mc else None), (mc["guidance"].cond if "guidance" in mc else None)


def calc_cond_uncond_batch(model, cond, uncond, x_in, timestep, model_options, cond_scale=1.0):
    """-> (cfg_result, cond_pred, uncond_pred).  Unlike the reference (:154) the CFG combine is fused in, because the
    per-half outputs only exist inside the combine kernel; both halves are still returned."""
    cctx = _single(cond, "cond")
    uctx = _single(uncond, "uncond") if uncond is not None else None
    return model.denoise_cfg(x_in, timestep, uctx, cctx, cond_scale, want_parts=True, transformer_options=model_options.get("transformer_options"),
                             control_model=cond[0].get("control"))


def sampling_function_inner(model, x, timestep, uncond, cond, cond_scale, model_options={}, seed=None, return_full=False):
    for k in _UNSUPPORTED_OPTS:
        if model_options.get(k):
            raise NotImplementedError(f"model_options['{k}'] is not supported by the native path")
    if math.isclose(cond_scale, 1.0) and not model_options.get("disable_cfg1_optimization", False):
        uncond_ = None  # :295-298
    else:
        uncond_ = uncond
    cfg_result, cond_pred, uncond_pred = calc_cond_uncond_batch(model, cond, uncond_, x, timestep, model_options, cond_scale)
    if return_full:
        return cfg_result, cond_pred, uncond_pred
    return cfg_result


def sampling_function(self, denoiser_params, cond_scale, cond_composition):
    """Same signature as the reference (:325): `self` is the CFGDenoiser."""
    unet_patcher = self.inner_model.inner_model.forge_objects.unet
    model = unet_patcher.model
    control = unet_patcher.controlnet_linked_list
    if unet_patcher.extra_concat_condition is not None:
        raise NotImplementedError("concat conditioning is outside the native hot path")
    if isinstance(denoiser_params.image_cond, torch.Tensor) and denoiser_params.image_cond.shape[1:] == denoiser_params.x.shape[1:] \
            and float(denoiser_params.image_cond.abs().max()) != 0.0:
        raise NotImplementedError("inpainting-model image conditioning is outside the native hot path")
    x, timestep = denoiser_params.x, denoiser_params.sigma
    uncond = compile_conditions(denoiser_params.text_uncond)
    cond = compile_weighted_conditions(denoiser_params.text_cond, cond_composition)
    if control is not None:  # :352-357
        for h in cond:
            h["control"] = control
        if uncond is not None:
            for h in uncond:
                h["control"] = control
    return sampling_function_inner(model, x, timestep, uncond, cond, cond_scale, unet_patcher.model_options,
                                   self.p.seeds[0], return_full=True)


def sampling_prepare(unet, x):
    """:366-399: the VRAM juggling (load_models_gpu) has nothing to do here -- weights are resident; what remains is handing every
    ControlNet the predictor and its sigma range (:393-397)."""
    real_model = unet.model
    for cnet in unet.list_controlnets():
        cnet.pre_run(real_model, lambda p: real_model.predictor.percent_to_sigma(p))


def sampling_cleanup(unet):
    for cnet in unet.list_controlnets():
        cnet.cleanup()
