"""Latent resize for the hires-fix pass -- what `torch.nn.functional.interpolate(samples, size, mode, antialias)` computes at
modules/processing.py:1459 for the modes of shared.latent_upscale_modes (modules/shared.py:55-63: bilinear, bicubic, nearest,
nearest-exact, with and without antialias), align_corners=False.

Every one of those modes is separable and linear: out[o] = sum_k w[o, k] * in[start[o] + k] per axis.  The (start, weights) tables
are a few hundred host floats built here the way ATen builds them (area_pixel source index for the plain modes; the
filter-support walk of the antialias kernels); the latent-sized work is ONE pass of fmx_resize_separable_f32 over the output."""
import numpy as np
import torch

from . import shared
from .. import hipops as ops

latent_upscale_default_mode = "Latent"
latent_upscale_modes = {
    "Latent": {"mode": "bilinear", "antialias": False},
    "Latent (antialiased)": {"mode": "bilinear", "antialias": True},
    "Latent (bicubic)": {"mode": "bicubic", "antialias": False},
    "Latent (bicubic antialiased)": {"mode": "bicubic", "antialias": True},
    "Latent (nearest)": {"mode": "nearest", "antialias": False},
    "Latent (nearest-exact)": {"mode": "nearest-exact", "antialias": False},
}
shared.latent_upscale_modes = latent_upscale_modes
shared.latent_upscale_default_mode = latent_upscale_default_mode


F = np.float32  # ATen does this index / weight arithmetic in the tensor's scalar type; fp32 here, so that ties round the same way


def _cubic(x, a):
    x, a = F(abs(x)), F(a)
    if x < 1.0:
        return ((a + F(2)) * x - (a + F(3))) * x * x + F(1)
    if x < 2.0:
        return (((x - F(5)) * x + F(8)) * x - F(4)) * a
    return F(0)


def axis_table(n_in, n_out, mode, antialias):
    """-> (start [n_out] int32, weights [n_out, K] fp32): out[o] = sum_k weights[o, k] * in[start[o] + k] along one axis.
    Border taps are folded together (index clamping), so start[o] + k is always in range."""
    scale = F(n_in) / F(n_out)
    rows = []
    for o in range(n_out):
        taps = {}

        def add(i, w):
            i = min(max(int(i), 0), n_in - 1)
            taps[i] = F(taps.get(i, F(0)) + F(w))
        if antialias and mode in ("bilinear", "bicubic"):
            half = F(1) if mode == "bilinear" else F(2)
            support = half * max(scale, F(1))
            inv = F(1) / max(scale, F(1))
            center = scale * F(o + 0.5)
            lo = max(int(center - support + F(0.5)), 0)
            hi = min(int(center + support + F(0.5)), n_in)
            f = (lambda v: max(F(1) - abs(v), F(0))) if mode == "bilinear" else (lambda v: _cubic(v, -0.5))
            ws = [f(F(j + lo - center + F(0.5)) * inv) for j in range(hi - lo)]
            total = F(sum(ws, F(0)))
            for j, w in enumerate(ws):
                add(lo + j, w / total if total != 0.0 else F(0))
        elif mode == "nearest":
            add(min(int(np.floor(F(o) * scale)), n_in - 1), 1.0)
        elif mode == "nearest-exact":
            add(min(int(np.floor(F(o + 0.5) * scale)), n_in - 1), 1.0)
        elif mode == "bilinear":
            real = max(scale * F(o + 0.5) - F(0.5), F(0))
            i0 = min(int(real), n_in - 1)
            lam = min(max(real - F(i0), F(0)), F(1))
            add(i0, F(1) - lam)
            add(i0 + 1, lam)
        elif mode == "bicubic":
            real = scale * F(o + 0.5) - F(0.5)
            i0 = int(np.floor(real))
            t = real - F(i0)
            for k in range(-1, 3):
                add(i0 + k, _cubic(F(k) - t, -0.75))
        else:
            raise ValueError(f"unknown latent scale mode {mode}")
        lo = min(taps)
        rows.append((lo, [float(taps.get(lo + k, 0.0)) for k in range(max(taps) - lo + 1)]))
    k = max(len(w) for _, w in rows)
    start = torch.tensor([min(lo, n_in - k) for lo, _ in rows], dtype=torch.int32)
    weights = torch.zeros(n_out, k, dtype=torch.float32)
    for o, (lo, w) in enumerate(rows):
        off = lo - int(start[o])
        weights[o, off:off + len(w)] = torch.tensor(w, dtype=torch.float32)
    return start, weights


def interpolate(samples, size, mode="bilinear", antialias=False):
    """fp32 NCHW [B, C, H, W] -> [B, C, size[0], size[1]]."""
    h, w = samples.shape[-2:]
    ys, yw = axis_table(h, int(size[0]), mode, antialias)
    xs, xw = axis_table(w, int(size[1]), mode, antialias)
    return ops.resize_separable(samples, ys, yw, xs, xw)
