"""bench.py's multi-rank contract without a GPU: `--gpus 2` with no launcher in the environment makes bench.py start its own two ranks
(torch.distributed.run on 127.0.0.1), the ranks form ONE process group of exactly N members (gloo here, RCCL on the GPU node), rank 0 prints
one JSON line with n_gpus = N and the number of ranks a collective actually saw; a launcher that started a different number of ranks than
--gpus is refused.  `--stub-engine` swaps the workload for a few CPU flops -- this tests the launcher, not the kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-engine"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_in_collective"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["launcher"] == "self"
    assert d["data"] == "stub" and d["scaling"] == "weak" and d["higher_is_better"] is True


def test_external_launcher_is_honoured_and_checked():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH]
    r = subprocess.run(base + ["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub-engine"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_in_collective"] == 2 and d["launcher"] == "external"
    # two ranks started, --gpus 4 claimed: no number may be printed
    r = subprocess.run(base + ["--gpus", "4", "--steps", "2", "--warmup", "0", "--stub-engine"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_process_default():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--stub-engine"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_in_collective"] == 1


def test_named_workloads_cover_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    assert a.gpus == 1 and a.config == "sdxl-b8-euler20"
    assert set(bench.CONFIGS) == {"sdxl-b8-euler20", "sd15-b4-eulera", "sdxl-b8-dpmpp2m30-vae", "flux-b2-bf16"}
    assert bench.CONFIGS["sd15-b4-eulera"]["sampler"] == "Euler a" and bench.CONFIGS["sd15-b4-eulera"]["batch"] == 4
    assert bench.CONFIGS["sdxl-b8-dpmpp2m30-vae"]["sampler"] == "DPM++ 2M" and bench.CONFIGS["sdxl-b8-dpmpp2m30-vae"]["nominal_steps"] == 30
    assert bench.CONFIGS["flux-b2-bf16"]["dtype"] == "bf16"
