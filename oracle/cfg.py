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
