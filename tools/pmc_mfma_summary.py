"""Summarise rocprofv3 --pmc passes of tools/pmc_kernels.py (tools/gpu_round2.sh pmc_mfma): per kernel and problem, matrix-pipe and VALU
utilisation from the SQ counters.

    python tools/pmc_mfma_summary.py gpurun_out/<tag> > profiles/<tag>_pmc_mfma_summary.json

Definitions (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs (per-XCD busy cycles = value / 8 = kernel duration in shader
cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (= 32 x number of v_mfma_f32_32x32x16 issued, checked against SQ_INSTS_MFMA);
SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves.
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs)      -- the "MFMA utilisation" at the clock the chip ran at
  valu_issue_frac  = 4 x SQ_ACTIVE_INST_VALU / the same denominator                     -- VALU (incl. MFMA issue) instruction-issue time
  valu_per_mfma    = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA                    -- non-matrix VALU instructions per MFMA
  clock_ghz        = kernel cycles / kernel duration (End - Start timestamps of the dispatch)
"""
import collections
import csv
import json
import os
import sys


def short(name):
    for key in ("attn_ws_kernel", "attn512_kernel", "conv3x3_gn_patch_kernel", "attn_short2_kernel", "attn_short_kernel", "attn_q64v3_kernel", "attn_q64v2_kernel", "attn_q64_kernel", "attn_kernel", "gemm256p_kernel", "gemm_kernel", "gn_apply_kernel", "gn_stats_kernel", "gn_finalize_kernel"):
        if key in name:
            i = name.index(key)
            j = name.find("(", i)
            return name[i:j if j > 0 else None].replace("(anonymous namespace)::", "")
    return None


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for sub in sorted(os.listdir(root)):
        f = os.path.join(root, sub, "pmc_counter_collection.csv")
        if sub not in ("pmc_mfma", "pmc_waves", "pmc_lds") or not os.path.exists(f):   # (the FETCH / WRITE passes profile another workload: bench.py)
            continue
        rows = list(csv.DictReader(open(f)))
        # the 8-wave GEMM kernels are persistent since round 3 (grid = one workgroup per CU whatever the problem), so the grid no longer tells
        # the shapes of tools/pmc_kernels.py apart: the workload launches each shape REPS times in a row -> number the runs of equal kernel names
        run_of, last, run = {}, None, -1
        for r in sorted({(int(x["Dispatch_Id"]), x["Kernel_Name"]) for x in rows}):
            if r[1] != last:
                run, last = run + 1, r[1]
            run_of[r[0]] = run
        for r in rows:
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            key = f"{k} grid={int(r['Grid_Size']) // int(r['Workgroup_Size'])}x{r['Workgroup_Size']} vgpr={r['VGPR_Count']} launch-group={run_of[int(r['Dispatch_Id'])]}"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key][r["Dispatch_Id"] + sub] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out = {}
    for key, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        o = {"dispatches_averaged": len(next(iter(d.values()))), "kernel_us_profiled": round(1e6 * sum(dur[key].values()) / len(dur[key]), 1)}
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            simd = cyc * 256 * 4
            o["kernel_cycles"] = round(cyc)
            o["clock_ghz"] = round(cyc / (o["kernel_us_profiled"] * 1e-6) / 1e9, 2)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                o["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4)
            if "SQ_ACTIVE_INST_VALU" in m:
                o["valu_issue_frac"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / simd, 4)
        if m.get("SQ_INSTS_MFMA"):
            o["mfma_insts"] = round(m["SQ_INSTS_MFMA"])
            o["valu_per_mfma"] = round((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"], 2)
        if m.get("SQ_WAVE_CYCLES"):
            for c, n in (("SQ_ACTIVE_INST_ANY", "wave_time_issuing"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled"), ("SQ_WAIT_ANY", "wave_time_waitcnt_or_barrier")):
                if c in m:
                    o[n] = round(m[c] / m["SQ_WAVE_CYCLES"], 3)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_bank_conflict_over_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
            if "GRBM_GUI_ACTIVE" in m or "kernel_cycles" in o:
                pass
        if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in m:
            o["wave_time_waiting_on_lds"] = round(m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], 3)
        o["raw"] = {c: round(v) for c, v in m.items()}
        out[key] = o
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    out["kernel_source_hash"] = kernel_source_hash()   # which kernel sources these counters were taken on
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
