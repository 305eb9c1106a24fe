"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpu_round.sh pmc) per kernel family -> JSON on stdout.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md
section HBM), so `fetch_bytes_corrected` = 2 * raw.  WRITE_SIZE is reported raw (uncalibrated)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            fam = "gemm256" if "gemm256" in name else "gemm" if "gemm_kernel" in name else "attn_short" if "attn_short" in name else "attn_q64v2" if "attn_q64v2" in name else \
                "attn_q64" if "attn_q64" in name else "attn" if "attn_kernel" in name else "gn_apply" if "gn_apply" in name else \
                "gn_stats" if "gn_stats" in name else "gn_finalize" if "gn_finalize" in name else "layernorm" if "ln_kernel" in name else None
            if fam is None:
                continue
            d = out.setdefault(fam, {"launches_FETCH_SIZE": 0, "launches_WRITE_SIZE": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
            d[f"launches_{counter}"] += 1
            d[f"{counter}_KiB"] += float(row["Counter_Value"])
for fam, d in out.items():
    nf, nw = max(1, d["launches_FETCH_SIZE"]), max(1, d["launches_WRITE_SIZE"])
    d["fetch_bytes_per_launch_raw"] = d["FETCH_SIZE_KiB"] * 1024 / nf
    d["fetch_bytes_per_launch_corrected"] = 2 * d["fetch_bytes_per_launch_raw"]
    d["write_bytes_per_launch_raw"] = d["WRITE_SIZE_KiB"] * 1024 / nw
    d["hbm_bytes_per_launch"] = d["fetch_bytes_per_launch_corrected"] + d["write_bytes_per_launch_raw"]
# which kernel sources these counters belong to: bench.py quotes `roofline.traffic` from this file only when the hash matches the binary it times
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402
out["kernel_source_hash"] = kernel_source_hash()
print(json.dumps(out, indent=1))
