"""Summary of a parity log (FMX_PARITY_LOG of the GPU suite): how many comparisons, where they sit against the reference's own fp16 floor, and the
headline rows with the unclamped per-pixel statistics.    python tools/parity_summary.py profiles/<tag>_parity_vs_fp16_floor.jsonl"""
import json
import sys

rows = [json.loads(ln) for ln in open(sys.argv[1])]
fl = [r for r in rows if "floor" in r]
ratio = sorted(((r["rms_rel"] / r["floor"]["rms_rel"], r["name"]) for r in fl if r["floor"]["rms_rel"] > 0), reverse=True)
med = ratio[len(ratio) // 2][0]
print(json.dumps({"comparisons": len(rows), "against_a_floor": len(fl), "pp_rel_above_1e-3": sum(r.get("pp_rel", 0.0) > 1e-3 for r in rows),
                  "rms_below_floor": sum(x < 1.0 for x, _ in ratio), "rms_ratio_median": round(med, 3), "rms_ratio_worst": round(ratio[0][0], 3),
                  "rows_above_floor_in_rms": [(round(x, 3), n) for x, n in ratio if x >= 1.0]}, indent=1))
keys = ("SDXL 1024x1024 batch 8, eight", "SDXL 1024x1024 30-step", "SDXL unet forward at full size (128x128", "SD1.5 512x512 20-step", "SD1.5 512x512 batch 4",
        "SDXL VAE decode 1024x1024 (sdxl_vae1024.pt) every", "SDXL VAE decode 1024x1024 (sdxl_config3_decode.pt) every", "flux forward at width 3072 with depth", "flux forward at width 3072 (24")
for r in rows:
    if any(r["name"].startswith(k) for k in keys):
        f = r.get("floor", {})
        print("%-110s pp_rel %.2e (floor %.2e)  rms %.2e (floor %.2e)  unclamped: %.1f %% of elements above 1e-3, 99.9th percentile %.2e" % (
            r["name"][:110], r["pp_rel"], f.get("pp_rel", float("nan")), r["rms_rel"], f.get("rms_rel", float("nan")), 100 * r.get("frac_gt_1e-3", float("nan")),
            r.get("pp_unclamped_p999", float("nan"))))
