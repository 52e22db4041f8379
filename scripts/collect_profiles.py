#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun session into profiles/ (tracked):
   profiles/<round>_<tag>_bench.json, _bench_kernel_stats.csv, _kv_mfma_pmc.json; refreshes profiles/kv_pmc_current.json
Usage: python scripts/collect_profiles.py r01 s5"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd, tag = sys.argv[1], sys.argv[2]
src = f"gpurun_out/{tag}"
os.makedirs("profiles", exist_ok=True)
if os.path.exists(f"{src}/bench.json") and os.path.getsize(f"{src}/bench.json"):
    shutil.copy(f"{src}/bench.json", f"profiles/{rnd}_{tag}_bench.json")
for f in glob.glob(f"{src}/prof/**/*kernel_stats*.csv", recursive=True)[:1]:
    shutil.copy(f, f"profiles/{rnd}_{tag}_bench_kernel_stats.csv")
out = {}
kname = None
for name in ("mfma", "insts", "fetch", "write"):
    for f in glob.glob(f"{src}/pmc_{name}/**/*counter_collection*.csv", recursive=True)[:1]:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kv_mfma" in r["Kernel_Name"] or "kv_gram" in r["Kernel_Name"]:
                kname = r["Kernel_Name"]
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
if out:
    doc = {
        "kernel": kname,
        "workload": f"fused K*V, RBF, n={os.environ.get('KV_ONLY_N', '500000')}, d=3, t=65 (scripts/kv_only.py)",
        "shape": [int(os.environ.get("KV_ONLY_N", "500000")), 3, 65],
        "counters": out,
        "notes": "separate rocprofv3 --pmc passes (scripts/gpu_session.sh); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* over all "
                 "SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE "
                 "under-counts wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)",
    }
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        doc["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"]["mean_per_launch"] + out["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    json.dump(doc, open(f"profiles/{rnd}_{tag}_kv_mfma_pmc.json", "w"), indent=1)
    json.dump(doc, open("profiles/kv_pmc_current.json", "w"), indent=1)
print(sorted(os.listdir("profiles")))
