import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # the CPU oracles (torch fp32 / fp64 on the host) are synchronisation-bound long before a GPU box's 128 cores: one SDXL forward takes 9.6 s on 16 threads
    # and 31 s on all 128 (profiles/r14b_cpu_baseline_thread_count.jsonl).  The layer-wise sharp tests evaluate several of those inside the GPU lease.
    try:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)
    except Exception:  # noqa: BLE001
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
