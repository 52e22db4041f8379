#!/bin/bash
# Evidence for the float64 path (round 4): the MEASURED float64 MFMA rate of the chip, the fused kernel's timings for d <= 16, rocprofv3 kernel
# stats and PMC passes (separate runs, no trace domains) at n = 100 000, t = 65 and t = 1.
# Usage: gpurun -- 'bash scripts/gpu_profile_f64.sh <tag>'  ->  gpurun_out/<tag>/...
set +e
TAG=${1:-f64}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 scripts/micro/mfma_f64_rate > $OUT/mfma_f64_rate.txt 2>&1; echo "micro rc=$?"; cat $OUT/mfma_f64_rate.txt
timeout 300 python scripts/f64_timing.py > $OUT/f64_timing.log 2>&1; echo "timing rc=$?"; tail -9 $OUT/f64_timing.log; cp gpurun_out/f64_timing.json $OUT/ 2>/dev/null
for t in 65 1; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_t$t -o f64 -- python $R/scripts/kv_f64_only.py 100000 3 $t 5 > $R/$OUT/stats_t$t.log 2>&1); echo "stats t=$t rc=$?"
  find $OUT/stats_t$t -name "*kernel_stats*.csv" | head -1 | xargs -r head -4 | cut -c1-220
  pass() { name=$1; shift; (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_t${t}_$name -o pmc -- python $R/scripts/kv_f64_only.py 100000 3 $t 2 > $R/$OUT/pmc_t${t}_$name.log 2>&1); echo "pmc t=$t $name rc=$?"; }
  pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
  pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  pass f64ops SQ_INSTS_VALU_MFMA_MOPS_F64
  pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
done
find $OUT -name "*kernel_trace*" -size +5M -delete
python - <<PY
import csv, glob, json, os
out = {}
for d in sorted(glob.glob("$OUT/pmc_t*")):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "kv_f64_kernel" in row.get("Kernel_Name", ""):
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        out[tag] = {k: sum(v) / max(1, len(set(range(len(v))))) for k, v in acc.items()}   # sum over dispatch rows / launches handled below
        out[tag + "_rows"] = {k: len(v) for k, v in acc.items()}
json.dump(out, open("$OUT/pmc_summary_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
