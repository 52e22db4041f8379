#!/bin/bash
set +e
OUT=gpurun_out/r2s16; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 scripts/micro/split_rates > $OUT/split_rates.txt 2>&1; echo "rates rc=$?"; cat $OUT/split_rates.txt
for ni in 2 4; do
  GPAMD_KGH_NI=$ni timeout 300 python scripts/kv_split_time.py r2s16_ni$ni 500000 64,65 > $OUT/time_ni$ni.log 2>&1; echo "ni=$ni rc=$?"; grep -E "^\{|Error|error" $OUT/time_ni$ni.log | cut -c1-400
done
pmc() { name=$1; shift; (cd /tmp && GPAMD_KV_SPLIT=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_$name -o pmc -- python $R/scripts/kv_only.py 500000 64 2 > $R/$OUT/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc insts SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r2s16/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "gramh" in k or "vsplit" in k or "reduce" in k:
                print(d.split("/")[-2], k, {c: round(x / max(1, cnt[(k, c)])) for c, x in v.items()})
PY
find $OUT -name "*kernel_trace*" -size +5M -delete
