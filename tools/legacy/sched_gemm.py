"""A/B of the DMA issue schedules of the 256x256 kernel (FMX_SCHED, -DFMX_ABLATE builds); checks results too."""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import forge_amd  # noqa
    from forge_amd import hipops as ops
    from tools.bench_kernels import timeit, rnd
    for rep in range(2):
        for (m, n, k) in [(8192, 8192, 8192), (4096, 4096, 4096), (16384, 10240, 1280), (16384, 2560, 1280), (65536, 640, 2560)]:
            x, w = rnd(m, k), rnd(n, k, scale=k ** -0.5)
            out = torch.empty(m, n, dtype=torch.float16, device="cuda")
            t = timeit(lambda: ops.conv_gemm(x, w, n, out=out, ld_out=n, force_tile=4), iters=10)
            err = float((out[:256].float() - x[:256].float() @ w.float().t()).abs().max())
            print(json.dumps({"sched": int(os.environ.get("FMX_SCHED", "0")), "m": m, "n": n, "k": k, "us": round(t * 1e6, 1), "tflops": round(2 * m * n * k / t / 1e12, 1), "maxerr": round(err, 4)}), flush=True)
else:
    for sc in (0, 1, 2, 3):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, FMX_SCHED=str(sc)))
