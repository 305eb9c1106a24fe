"""Effective shader clock and cycles per K-tile of the pipelined 256x256 GEMM under SUSTAINED load
(stamping build made OUTSIDE csrc/: `python tools/patch_clock_stamps.py clock && tools/build_patched_file.sh clock tools/_build/src/clock_clock/fmx_gemm256p.hip`,
selected with FMX_ALLOW_KNOBS=1 FMX_LIB=tools/_build/libfmx_clock.so: workgroup 8 writes its K-loop s_memtime / s_memrealtime deltas into its output tile)."""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops

TILE = int(os.environ.get("FMX_TILE", "6"))
BN = 320 if TILE == 7 else 256
RES = os.environ.get("FMX_CLOCK_RESIDUAL", "0") == "1"   # round 3: the residual-adding form (out = x + f(x) in place), as attn to_out / ff.net.2
for (m, n, k, reps, act) in [(16384, 1280, 1280, 200, 0), (16384, 10240, 1280, 60, 1), (65536, 640, 640, 200, 0), (16384, 1280, 5120, 100, 0)]:
    if RES and act:
        continue
    for data in ("randn",):
        x = (torch.randn(m, k, device="cuda") if data == "randn" else torch.zeros(m, k, device="cuda")).half()
        w = (torch.randn(n, k, device="cuda") * k ** -0.5 if data == "randn" else torch.zeros(n, k, device="cuda")).half()
        out = torch.zeros(m, n // 2 if act else n, dtype=torch.float16, device="cuda")
        if RES:
            out.normal_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.conv_gemm(x, w, n, out=out, ld_out=out.shape[1], act=act, force_tile=TILE, residual=out if RES else None)
        s.record()
        for _ in range(reps):
            ops.conv_gemm(x, w, n, out=out, ld_out=out.shape[1], act=act, force_tile=TILE, residual=out if RES else None)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / reps * 1e-3
        words = out.view(-1).view(torch.int32)
        # find the marker
        idx = (words == 0x5eed5eed).nonzero()
        if idx.numel() == 0:
            print("marker not found"); continue
        i = int(idx[0]) - 3
        cyc, rt, kt = [int(v) & 0xffffffff for v in words[i:i + 3].tolist()]
        pro, epi = [int(v) & 0xffffffff for v in words[i + 4:i + 6].tolist()]
        tiles = -(-m // 256) * -(-n // BN)
        rounds = -(-tiles // 256)
        print(f"   per workgroup: prologue {pro * 10} ns, K loop {rt * 10} ns, epilogue+store drain {epi * 10} ns; kernel wall {t * 1e6:.1f} us over {rounds} round(s) "
              f"-> {t * 1e6 / rounds:.1f} us per round vs {(pro + rt + epi) * 0.01:.1f} us inside the workgroup")
        print(f"{m}x{n}x{k} act={act} residual={int(RES)} {data}: {2*m*n*k/t/1e12:.0f} TF/s wall; K loop {cyc} cycles / {rt} ticks -> {cyc/kt:.0f} cycles per K-tile, "
              f"clock {cyc/rt*0.1:.3f} GHz, MFMA-pipe busy {8 * BN * kt / cyc * 100:.0f}% "
              f"(ideal {8 * BN} cycles per K-tile: 256 x {BN} x 64 x 2 FLOP / 4096 per cycle and CU)", flush=True)
