"""Timing ablations of the 256x256 GEMM kernel (library built with EXTRA=-DFMX_ABLATE; results are garbage for ABL != 0).
One process per ABL value because the library reads FMX_ABL once."""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import forge_amd  # noqa
    from forge_amd import hipops as ops
    from tools.bench_kernels import timeit, rnd
    for (m, n, k) in [(8192, 8192, 8192), (4096, 4096, 4096), (16384, 10240, 1280), (16384, 2560, 1280)]:
        x, w = rnd(m, k), rnd(n, k, scale=k ** -0.5)
        out = torch.empty(m, n, dtype=torch.float16, device="cuda")
        t = timeit(lambda: ops.conv_gemm(x, w, n, out=out, ld_out=n, force_tile=4), iters=10)
        print(json.dumps({"abl": int(os.environ.get("FMX_ABL", "0")), "m": m, "n": n, "k": k, "us": round(t * 1e6, 1), "tflops": round(2 * m * n * k / t / 1e12, 1)}), flush=True)
else:
    for abl in [int(a) for a in os.environ.get("ABLS", "0,32,64,38,70,6").split(",")]:
        env = dict(os.environ, FMX_ABL=str(abl))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
