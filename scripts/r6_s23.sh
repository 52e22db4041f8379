#!/bin/bash
# round 6, GPU session 23: protein-shaped closure on the final binary -- phases, plain timing, rocprofv3 kernel statistics (where the 20 ms go now)
set +e
OUT=gpurun_out/r6s23; mkdir -p $OUT
export TMPDIR=/tmp
true
true
python - <<'PY'
import json
d = json.load(open("gpurun_out/workload_breakdown_protein_plain.json")); print(d["stages"][0]["seconds_per_iteration"])
d = json.load(open("gpurun_out/workload_breakdown_protein_phases.json")); st = d["stages"][0]
print(json.dumps(st["phases_exclusive_seconds_per_iteration"], indent=1)); print(st["calls_per_iteration"])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_protein -o prof -- python $GRAFT_REPO_ROOT/scripts/workload_breakdown.py protein plain 6 > $GRAFT_REPO_ROOT/$OUT/3_protein_rocprof.log 2>&1); echo "[protein plain under rocprof] rc=$?"
cp gpurun_out/workload_breakdown_protein_*.json $OUT/ 2>/dev/null
find $OUT/prof_protein -name "*kernel_stats.csv" -exec cp {} $OUT/protein_kernel_stats.csv \;
rm -rf $OUT/prof_protein
head -40 $OUT/protein_kernel_stats.csv | cut -c1-200
