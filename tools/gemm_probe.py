"""Probe: 256x256 kernel (tile 4) vs 128x128 (tile 1) on power-of-two vs padded K strides, and a few real shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import timeit, rnd

shapes = [(8192, 8192, 8192), (8192, 8192, 8256), (8192, 8192, 8128), (4096, 4096, 4096), (4096, 4096, 4160), (16384, 10240, 1280), (16384, 10240, 1344),
          (16384, 2560, 1280), (16384, 1280, 5120), (16384, 1280, 5184), (65536, 640, 2560), (65536, 5120, 640)]
for (m, n, k) in shapes:
    x, w = rnd(m, k), rnd(n, k, scale=k ** -0.5)
    out = torch.empty(m, n, dtype=torch.float16, device="cuda")
    for tile in (1, 4):
        t = timeit(lambda: ops.conv_gemm(x, w, n, out=out, ld_out=n, force_tile=tile), iters=10)
        print(json.dumps({"tile": tile, "m": m, "n": n, "k": k, "us": round(t * 1e6, 1), "tflops": round(2 * m * n * k / t / 1e12, 1)}), flush=True)
