"""Timeline of ONE persistent workgroup (blockIdx 8) of the 8-wave GEMM kernel over the output tiles it walks, under sustained load: for every tile the
100 MHz real-time counter at the tile's top, in front of the K loop, behind the first K-tile's barrier, behind the K loop, behind the epilogue
(library built by tools/build_timeline.sh; select it with FMX_ALLOW_KNOBS=1 FMX_LIB=tools/_build/libfmx_timeline.so).  What is left between two tiles, and where.

    FMX_ALLOW_KNOBS=1 FMX_LIB=tools/_build/libfmx_timeline.so [FMX_GEMM_XTILE=0] python tools/tile_timeline.py
"""
import json
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from forge_amd.backend.nn.unet import _fold_layernorm

CUS = torch.cuda.get_device_properties(0).multi_processor_count & ~7


def xcd_remap(orig, nwg):
    q, r = nwg >> 3, nwg & 7
    xcd = orig & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + (orig >> 3)


def tile_origin(lid, tiles_m, tiles_n):
    wg = xcd_remap(lid, tiles_m * tiles_n)
    per_group = 8 * tiles_n
    grp = wg // per_group
    first_m = grp * 8
    gsz = min(8, tiles_m - first_m)
    in_g = wg - grp * per_group
    tn = in_g // gsz
    return first_m + (in_g - tn * gsz), tn


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).half()


def run(name, m, n, k, reps, geglu, ln):
    x, w, b = rnd(m, k), rnd(n, k, scale=k ** -0.5), rnd(n)
    kw = {}
    if geglu:
        w, b = ops.geglu_interleave(w, b)
        kw["act"] = ops.ACT_GEGLU
    if ln:   # the ff.net.0 / q|k form: LayerNorm folded, statistics from a producer launch
        rs = ops.RowStats(m, k)
        h = rnd(m, k)
        ops.linear(rnd(m, k), rnd(k, k, scale=k ** -0.5), rnd(k), residual=h, out=h, ld_out=k, row_stats=rs, force_tile=7)
        assert rs.parts > 0
        w, cs, b = _fold_layernorm(w, b, 1 + 0.2 * rnd(k), 0.1 * rnd(k))
        kw["ln"] = (rs, cs, 1e-5)
        x = h
    cols = n // 2 if geglu else n
    out = torch.zeros(m, cols, dtype=torch.float16, device="cuda")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=cols, force_tile=0 if ln else 7, **kw)
    s.record()
    for _ in range(reps):
        ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=cols, force_tile=0 if ln else 7, **kw)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    tiles_m, tiles_n = -(-m // 256), -(-n // 320)
    nwg = tiles_m * tiles_n
    grid = min(nwg, CUS)
    words = out.view(torch.int32)
    rows = []
    lid = 8
    while lid < nwg:
        tm, tn = tile_origin(lid, tiles_m, tiles_n)
        col = (tn * 320 // 2 if geglu else tn * 320) // 2      # int32 word index inside the row
        v = [int(t) & 0xffffffff for t in words[tm * 256, col:col + 7].tolist()]
        if v[5] != 0x71e11e00 or v[6] != lid:
            print(f"   {name}: no stamp at tile {lid} ({tm}, {tn}): {v}")
            return
        rows.append(v[:5])
        lid += grid
    kt = k // 64
    print(f"{name}: M={m} N={n} K={k} ({kt} K-tiles), {len(rows)} tiles per workgroup, launch {us:.1f} us wall ({2 * m * n * k / us / 1e6:.0f} TFLOP/s), "
          f"xtile={os.environ.get('FMX_GEMM_XTILE', '1')}; times in us since the workgroup's entry")
    out_rows = []
    for i, (top, k0, fb, k1, e1) in enumerate(rows):
        nxt = rows[i + 1][0] if i + 1 < len(rows) else None
        d = {"tile": i, "top": top / 100, "prologue": (k0 - top) / 100, "first_k_tile": (fb - k0) / 100, "other_k_tiles_each": round((k1 - fb) / 100 / max(kt - 1, 1), 3),
             "k_loop": (k1 - k0) / 100, "epilogue": (e1 - k1) / 100, "to_next_top": None if nxt is None else (nxt - e1) / 100, "tile_total": None if nxt is None else (nxt - top) / 100}
        out_rows.append(d)
        print("   " + json.dumps(d))
    return out_rows


if __name__ == "__main__":
    run("ff.net.0 (LayerNorm consumer, GEGLU)", 16384, 10240, 1280, 60, True, True)
    run("plain GEGLU", 16384, 10240, 1280, 60, True, False)
    run("GEGLU at 640", 65536, 5120, 640, 60, True, False)
    run("q|k (LayerNorm consumer)", 16384, 2560, 1280, 120, False, True)
    run("plain, K = 640", 65536, 640, 640, 120, False, False)
