#!/bin/bash
# bench line (fp32 headline + split block), rocprofv3 kernel stats of the same command, PMC passes of the split kernel at t = 65
set +e
OUT=gpurun_out/r2s24; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-extras --skip-parity > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-220; done
pmc() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_$name -o pmc -- python $R/scripts/kv_only.py 500000 65 2 > $R/$OUT/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
find $OUT -name "*kernel_trace*" -size +5M -delete
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("gpurun_out/r2s24/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "gramh" in k or "vsplit" in k:
                out.setdefault(k, {}).update({c: round(x / max(1, cnt[(k, c)])) for c, x in v.items()})
json.dump(out, open("gpurun_out/r2s24/pmc_split_t65.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
