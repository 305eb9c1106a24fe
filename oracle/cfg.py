"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the classifier-free-guidance batching of backend/sampling/sampling_function.py for the plain
txt2img case (one cond + one uncond per image, no areas/masks/controls):
  :154-289 calc_cond_uncond_batch -- cond and uncond packed into ONE UNet batch ordered [uncond ; cond]
           (to_run is popped from the back, :187-229), outputs weighted by mult=1 over counts 1+1e-37;
  :292-322 sampling_function_inner -- cond_scale == 1 skips the uncond half entirely; otherwise
           uncond + (cond - uncond) * cond_scale in fp32.
"""
import math

import torch


def _cat(a, b):
    if isinstance(a, dict):
        return {k: torch.cat([a[k], b[k]]) for k in a}
    return torch.cat([a, b])


def _ctx_y(c):
    if isinstance(c, dict):
        return c["crossattn"], c.get("vector")
    return c, None


def cfg_denoise(apply_model_fn, x, sigma, uncond, cond, cond_scale):
    """apply_model_fn(x[Bu], sigma[Bu], context[Bu,T,D], y|None) -> denoised[Bu]. Returns (cfg, cond_pred, uncond_pred)."""
    b = x.shape[0]
    if math.isclose(cond_scale, 1.0):
        ctx, y = _ctx_y(cond)
        cond_pred = apply_model_fn(x, sigma, ctx, y)
        cond_pred = cond_pred * 1.0 / (1.0 + 1e-37)
        uncond_pred = torch.zeros_like(x) / 1e-37  # sampling_function.py:155-159,284-288 (all-zero accumulators)
        return uncond_pred + (cond_pred - uncond_pred) * cond_scale, cond_pred, uncond_pred
    ctx, y = _ctx_y(_cat(uncond, cond))
    out = apply_model_fn(torch.cat([x, x]), torch.cat([sigma, sigma]), ctx, y)
    uncond_pred, cond_pred = out[:b], out[b:]
    cnt = torch.ones_like(x) * 1e-37 + 1.0
    cond_pred = cond_pred / cnt
    uncond_pred = uncond_pred / cnt
    return uncond_pred + (cond_pred - uncond_pred) * cond_scale, cond_pred, uncond_pred


def cfg_denoise_general(apply_model_fn, x, sigma, uncond, cond, composition, cond_scale, options=None):
    """The general form (sampling_function.py:154-322, condition.py:122-142): `composition[i]` = [(row of `cond`, weight), ...] per image
    (AND-composed prompts).  Part k of every image forms one full-batch cond entry with strength = the LAST image's weight for that part
    (condition.py:129-131); all entries + the uncond go through ONE model call stacked [uncond, part K-1, ..., part 0]; cond_pred is the
    strength-weighted average; CFG uses scale * edit_strength when the strengths do not sum to 1.  options: sampler_pre_cfg_function /
    sampler_cfg_function / sampler_post_cfg_function / model_function_wrapper as model_options carries them."""
    options = options or {}
    b = x.shape[0]
    idx = lambda t, rows: ({k: v[rows] for k, v in t.items()} if isinstance(t, dict) else t[rows])
    parts = []
    for part in zip(*composition):
        parts.append((idx(cond, [i for i, _ in part]), part[-1][1]))
    edit_strength = sum(w for _, w in parts)
    use_uncond = not math.isclose(cond_scale, 1.0)
    entries = ([(uncond, 1.0, 1)] if use_uncond else []) + [(c, w, 0) for c, w in reversed(parts)]
    stacked = entries[0][0]
    for e in entries[1:]:
        stacked = _cat(stacked, e[0])
    ctx, y = _ctx_y(stacked)
    n = len(entries)
    x_in, s_in = torch.cat([x] * n), torch.cat([sigma] * n)
    if "model_function_wrapper" in options:
        out = options["model_function_wrapper"](lambda xx, tt, **c: apply_model_fn(xx, tt, c["c_crossattn"], c.get("y")),
                                                {"input": x_in, "timestep": s_in, "c": {"c_crossattn": ctx, **({"y": y} if y is not None else {})},
                                                 "cond_or_uncond": [e[2] for e in entries]})
    else:
        out = apply_model_fn(x_in, s_in, ctx, y)
    chunks = out.chunk(n)
    acc = {0: (torch.zeros_like(x), torch.ones_like(x) * 1e-37), 1: (torch.zeros_like(x), torch.ones_like(x) * 1e-37)}
    for (c, w, kind), o in zip(entries, chunks):
        acc[kind] = (acc[kind][0] + o * w, acc[kind][1] + w)
    cond_pred, uncond_pred = acc[0][0] / acc[0][1], acc[1][0] / acc[1][1]
    if "sampler_cfg_function" in options:
        res = x - options["sampler_cfg_function"]({"cond": x - cond_pred, "uncond": x - uncond_pred, "cond_scale": cond_scale, "timestep": sigma,
                                                   "input": x, "sigma": sigma, "cond_denoised": cond_pred, "uncond_denoised": uncond_pred})
    elif not math.isclose(edit_strength, 1.0):
        res = uncond_pred + (cond_pred - uncond_pred) * cond_scale * edit_strength
    else:
        res = uncond_pred + (cond_pred - uncond_pred) * cond_scale
    for fn in options.get("sampler_post_cfg_function", []):
        res = fn({"denoised": res, "uncond_denoised": uncond_pred, "cond_denoised": cond_pred, "sigma": sigma, "input": x})
    return res
