#!/bin/bash
# round 6, GPU session 6: variational variance vs solve tolerance (bounded sweep); default-settings-vs-oracle test; protein breakdown + rocprof on the new
# pivoted-Cholesky / coefficient-sum kernels; clock / power with the repaired ablations; precond tests on the new sum kernel
set +e
OUT=gpurun_out/r6s6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python scripts/variational_variance_sweep.py > $OUT/1_variational_sweep.log 2>&1; echo "[variational sweep] rc=$?"; grep "precond_rank" $OUT/1_variational_sweep.log
timeout 300 python -m pytest tests/test_gpu_love_vs_oracle.py tests/test_gpu_bbmm.py -m gpu -q -k "default_settings or precond or pivoted or inv_quad" > $OUT/2_tests.log 2>&1; echo "[default settings vs oracle + bbmm] rc=$?"; tail -12 $OUT/2_tests.log
timeout 200 python scripts/workload_breakdown.py protein phases > $OUT/3_protein_phases.log 2>&1; echo "[protein phases] rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/workload_breakdown_protein_phases.json")); st = d["stages"][0]
print(st["seconds_per_iteration"], st["cg_iterations"])
for k, v in st["phases_exclusive_seconds_per_iteration"].items(): print("  %-70s %.2f ms" % (k, v * 1e3))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_protein -o prof -- python $GRAFT_REPO_ROOT/scripts/workload_breakdown.py protein plain 6 > $GRAFT_REPO_ROOT/$OUT/4_protein_plain.log 2>&1); echo "[protein plain under rocprof] rc=$?"
find $OUT/prof_protein -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-160
python - <<'PY'
import json
d = json.load(open("gpurun_out/workload_breakdown_protein_plain.json")); print(d["stages"][0]["seconds_per_iteration"], d["timed_seconds_total"])
PY
timeout 300 python scripts/kgh_clock_power.py r6 500000 3 > $OUT/5_clock_power.log 2>&1; echo "[clock / power] rc=$?"; grep -E "without|one block" $OUT/5_clock_power.log
cp gpurun_out/posterior_at_size_c2_variational.json gpurun_out/default_settings_vs_oracle.json gpurun_out/workload_breakdown_protein_*.json gpurun_out/kgh_clock_power_r6.json $OUT/ 2>/dev/null
find $OUT/prof_protein -name "*kernel_stats.csv" -exec cp {} $OUT/protein_kernel_stats.csv \;
rm -rf $OUT/prof_protein
