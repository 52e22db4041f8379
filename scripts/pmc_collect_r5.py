#!/usr/bin/env python3
"""rocprofv3 --pmc passes of scripts/r5_s9.sh -> one JSON per kernel under profiles/: python scripts/pmc_collect_r5.py <session dir> <out prefix>"""
import collections
import csv
import glob
import json
import sys

src, prefix = sys.argv[1], sys.argv[2]
for case, match in (("directh_t11", "kv_directh_kernel"), ("f64_t65", "kv_f64_kernel"), ("f64_t1", "kv_f64v_kernel")):
    out, kname = {}, None
    for f in glob.glob(f"{src}/pmc_{case}_*/**/*counter_collection*.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                kname = r["Kernel_Name"]
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if not out:
        continue
    der = {}
    if "GRBM_GUI_ACTIVE" in out:
        cyc = out["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
        der["gui_active_cycles_per_xcd"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in out:
            der["matrix_pipe_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0 / cyc
        if "SQ_ACTIVE_INST_VALU" in out:
            der["valu_busy_frac"] = 4.0 * out["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / 1024.0 / cyc
    if "SQ_INSTS_VALU" in out and out.get("SQ_INSTS_MFMA", {}).get("mean_per_launch"):
        der["valu_per_mfma"] = (out["SQ_INSTS_VALU"]["mean_per_launch"] - out["SQ_INSTS_MFMA"]["mean_per_launch"]) / out["SQ_INSTS_MFMA"]["mean_per_launch"]
    json.dump({"kernel": kname, "case": case, "counters": out, "derived": der,
               "notes": "separate rocprofv3 --pmc passes; GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_* over all SIMDs; SQ_ACTIVE_* count quad-cycles"},
              open(f"{prefix}_{case}.json", "w"), indent=1)
    print(case, kname, der)
