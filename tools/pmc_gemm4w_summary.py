"""Summarise the counter passes of tools/pmc_gemm4w.py (gpurun_out/<tag>/pmc4w_<pass>/pmc_counter_collection.csv) per kernel and problem.
    python tools/pmc_gemm4w_summary.py gpurun_out/<tag> > profiles/<tag>_pmc_gemm4w_vs_256x320.json
Conventions as tools/pmc_mfma_summary.py: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* cycle counters are per SIMD / per wave quad-cycles."""
import collections
import csv
import json
import os
import sys


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for sub in sorted(os.listdir(root)):
        f = os.path.join(root, sub, "pmc_counter_collection.csv")
        if not sub.startswith("pmc4w_") or not os.path.exists(f):
            continue
        rows = [r for r in csv.DictReader(open(f)) if "gemm4w_kernel" in r["Kernel_Name"] or "gemm256p_kernel" in r["Kernel_Name"]]
        # launch order of tools/pmc_gemm4w.py: (K = 1280: tile 7 x REPS, tile 10 x REPS), (K = 5120: the same) -> the run index of a kernel name tells K
        seen = collections.defaultdict(list)
        for did, name in sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in rows}):
            seen["4w" if "gemm4w" in name else "256x320"].append(did)
        kof = {}
        for kern, ids in seen.items():
            half = len(ids) // 2
            for i, did in enumerate(ids):
                kof[did] = (kern, 1280 if i < half else 5120)
        for r in rows:
            kern, k = kof[int(r["Dispatch_Id"])]
            key = f"{kern} M=16384 N=1280 K={k} grid={int(r['Grid_Size']) // int(r['Workgroup_Size'])}x{r['Workgroup_Size']} vgpr={r['VGPR_Count']} lds={r.get('LDS_Block_Size', '?')}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    out = {}
    for key, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        us = 1e6 * sum(dur[key]) / len(dur[key])
        o = {"kernel_us_profiled": round(us, 1)}
        k = int(key.split("K=")[1].split()[0])
        o["tflops_profiled"] = round(2 * 16384 * 1280 * k / (us * 1e-6) / 1e12, 1)
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            o["clock_ghz"] = round(cyc / (us * 1e-6) / 1e9, 2)
            simd = cyc * 256 * 4
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                o["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 3)
                o["mfma_busy_x_ghz"] = round(o["mfma_busy_frac"] * o["clock_ghz"], 3)
            if "SQ_ACTIVE_INST_VALU" in m:
                o["valu_issue_frac"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / simd, 3)
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
                if c in m:
                    o[c.lower() + "_per_wave_cycle"] = round(m[c] / m["SQ_WAVE_CYCLES"], 3)
        if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
            o["lds_bank_conflict_per_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum",
                  "TCP_TCC_READ_REQ_LATENCY_sum", "TA_TA_BUSY_sum", "TA_BUFFER_READ_LDS_WAVEFRONTS_sum", "FETCH_SIZE", "TCP_PENDING_STALL_CYCLES_sum", "SQ_LDS_IDX_ACTIVE",
                  "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            if c in m:
                o[c] = round(m[c])
        if "TCP_TCC_READ_REQ_LATENCY_sum" in m and m.get("TCP_TCC_READ_REQ_sum"):
            o["l2_read_latency_cycles"] = round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"], 1)
        if "TCC_HIT_sum" in m and (m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)):
            o["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)), 3)
        out[key] = o
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
