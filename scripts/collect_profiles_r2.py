#!/usr/bin/env python3
"""Summaries of a scripts/gpu_profile_r2.sh session -> profiles/ (tracked):
   profiles/r02_<tag>_bench_kernel_stats.csv, profiles/r02_<tag>_kv_pmc_t<t>.json, and profiles/kv_pmc_current.json (t = 65).
Usage: python scripts/collect_profiles_r2.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/{tag}"
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(f"{src}/prof/**/*kernel_stats*.csv", recursive=True)[:1]:
    shutil.copy(f, f"profiles/r02_{tag}_bench_kernel_stats.csv")
PEAK_TF, CLK = 157.3, 2.4e9
n = 500_000
for t in (65, 1, 11, 16, 17):
    out, kname = {}, None
    for name in ("mfma", "insts", "fetch", "write"):
        for f in glob.glob(f"{src}/pmc_t{t}_{name}/**/*counter_collection*.csv", recursive=True)[:1]:
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "gpamd::kv_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                    kname = r["Kernel_Name"]
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if not out:
        continue
    doc = {
        "kernel": kname,
        "workload": f"fused K*V, RBF, n={n}, d=3, t={t} (scripts/kv_only.py)",
        "shape": [n, 3, t],
        "counters": out,
        "notes": "separate rocprofv3 --pmc passes (scripts/gpu_profile_r2.sh); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* over all "
                 "SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE "
                 "under-counts wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)",
    }
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        doc["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"]["mean_per_launch"] + out["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in out:
        secs = out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0 / CLK          # per-XCD active cycles -> seconds (approx. launch time)
        doc["derived"] = {"launch_seconds_from_gui_active": secs}
        if "hbm_bytes_per_launch" in doc:
            doc["derived"]["hbm_GBps"] = doc["hbm_bytes_per_launch"] / secs / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in out:
            doc["derived"]["matrix_pipe_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0 / (out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
        if "SQ_ACTIVE_INST_VALU" in out:
            doc["derived"]["valu_busy_frac"] = 4.0 * out["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / 1024.0 / (out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
    json.dump(doc, open(f"profiles/r02_{tag}_kv_pmc_t{t}.json", "w"), indent=1)
    if t == 65:
        json.dump(doc, open("profiles/kv_pmc_current.json", "w"), indent=1)
print(sorted(f for f in os.listdir("profiles") if f.startswith("r02_")))
