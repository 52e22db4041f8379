#!/bin/bash
# round 6, GPU session 7: changed tests on the current binary (sum kernel, plan cache, variance tolerance factor, near-limit clouds, oracle in float32)
set +e
OUT=gpurun_out/r6s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_love_vs_oracle.py tests/test_gpu_recenter.py tests/test_gpu_dense_at_size.py tests/test_gpu_bbmm.py tests/test_gpu_multitask.py tests/test_gpu_model.py -m gpu -q > $OUT/1_tests.log 2>&1; echo "[tests] rc=$?"; tail -40 $OUT/1_tests.log
timeout 200 python scripts/workload_breakdown.py protein phases > $OUT/2_protein_phases.log 2>&1; echo "[protein phases] rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/workload_breakdown_protein_phases.json")); st = d["stages"][0]
print(st["seconds_per_iteration"], st["cg_iterations"])
for k, v in st["phases_exclusive_seconds_per_iteration"].items(): print("  %-70s %.2f ms" % (k, v * 1e3))
PY
timeout 200 python scripts/workload_breakdown.py protein plain 8 > $OUT/3_protein_plain.log 2>&1; echo "[protein plain] rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/workload_breakdown_protein_plain.json")); print(d["stages"][0]["seconds_per_iteration"])
PY
cp gpurun_out/default_settings_vs_oracle.json gpurun_out/posterior_at_size_*.json gpurun_out/workload_breakdown_protein_*.json $OUT/ 2>/dev/null
