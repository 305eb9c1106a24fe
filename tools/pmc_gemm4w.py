"""rocprofv3 --pmc workload for the round-4 A/B of the two GEMM kernels on the SAME problems: the 256x320 tile (one 8-wave workgroup per CU, force_tile 7)
against the 256x160 tile with two 4-wave workgroups per CU (force_tile 10), K = 1280 (one 256x320 tile per CU: the launch the overlap was meant for) and
K = 5120 (the K loop itself).  tools/gpu_round2.sh pmc_4w runs the counter passes, tools/pmc_gemm4w_summary.py reads them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import rnd

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (m, n, k) in [(16384, 1280, 1280), (16384, 1280, 5120)]:
    x, w, b = rnd(m, k), rnd(n, k, scale=k ** -0.5), rnd(n)
    out = torch.empty(m, n, dtype=torch.float16, device="cuda")
    for tile in (7, 10):
        for _ in range(REPS):
            ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=n, force_tile=tile)
torch.cuda.synchronize()
