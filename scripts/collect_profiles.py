#!/usr/bin/env python3
"""Summaries of a scripts/gpu_profile.sh session -> profiles/ (tracked): profiles/r<round>_<tag>_bench_kernel_stats.csv,
profiles/r<round>_<tag>_kv_pmc_<path>_t<t>.json, profiles/kv_pmc_current.json (fp32 contraction, t = 65) and
profiles/kv_pmc_split_current.json (split contraction, t = 65) -- the two files bench.py reads its `traffic` figures from.
Usage: python scripts/collect_profiles.py <tag> <round>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "03")
src = f"gpurun_out/{tag}"
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(f"{src}/prof/**/*kernel_stats*.csv", recursive=True)[:1]:
    shutil.copy(f, f"profiles/r{rnd}_{tag}_bench_kernel_stats.csv")
if os.path.exists(f"{src}/rocprof_bench.json"):
    shutil.copy(f"{src}/rocprof_bench.json", f"profiles/r{rnd}_{tag}_bench_under_rocprof.json")
n = 500_000
for path, t in (("f32", 65), ("split", 65), ("split", 1), ("split", 11)):
    out, kname = {}, None
    for name in ("mfma", "insts", "lds", "fetch", "write"):
        for f in glob.glob(f"{src}/pmc_{path}_t{t}_{name}/**/*counter_collection*.csv", recursive=True)[:1]:
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if "gpamd::kv_" in kn and "reduce" not in kn and "vsplit" not in kn:
                    kname = kn
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if not out:
        continue
    doc = {
        "kernel": kname,
        "workload": f"fused K*V, RBF, n={n}, d=3, t={t}, {path} contraction (scripts/kv_only.py)",
        "shape": [n, 3, t],
        "counters": out,
        "notes": "separate rocprofv3 --pmc passes (scripts/gpu_profile.sh); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* over all "
                 "SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE "
                 "under-counts wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)",
    }
    der = {}
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        der["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"]["mean_per_launch"] + out["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
        doc["hbm_bytes_per_launch"] = der["hbm_bytes_per_launch"]
    if "GRBM_GUI_ACTIVE" in out:
        cyc = out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0                    # per-XCD active cycles of one launch
        der["gui_active_cycles_per_xcd"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in out:
            der["matrix_pipe_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0 / cyc
        if "SQ_ACTIVE_INST_VALU" in out:
            der["valu_busy_frac"] = 4.0 * out["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / 1024.0 / cyc
        if "SQ_ACTIVE_INST_LDS" in out:
            der["lds_busy_frac"] = 4.0 * out["SQ_ACTIVE_INST_LDS"]["mean_per_launch"] / 1024.0 / cyc
    if "SQ_INSTS_VALU" in out and "SQ_INSTS_MFMA" in out and out["SQ_INSTS_MFMA"]["mean_per_launch"]:
        der["valu_per_mfma"] = (out["SQ_INSTS_VALU"]["mean_per_launch"] - out["SQ_INSTS_MFMA"]["mean_per_launch"]) / out["SQ_INSTS_MFMA"]["mean_per_launch"]
    if "SQ_LDS_BANK_CONFLICT" in out and "SQ_LDS_IDX_ACTIVE" in out and out["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]:
        der["lds_bank_conflict_frac"] = out["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / out["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]
    doc["_derived"] = der
    json.dump(doc, open(f"profiles/r{rnd}_{tag}_kv_pmc_{path}_t{t}.json", "w"), indent=1)
    if t == 65:
        json.dump(doc, open("profiles/kv_pmc_current.json" if path == "f32" else "profiles/kv_pmc_split_current.json", "w"), indent=1)
print(sorted(f for f in os.listdir("profiles") if f.startswith(f"r{rnd}_{tag}")))
