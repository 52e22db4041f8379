#!/bin/bash
# round 6, GPU session 31: preconditioner apply per (n, rank, columns), library before / after the projection-kernel changes (same box)
set +e
OUT=gpurun_out/r6s31; mkdir -p $OUT
GPAMD_LIBRARY=$GRAFT_REPO_ROOT/gpytorch_amd/csrc/tune/libgpamd_base.so timeout 300 python scripts/precond_apply_timing.py base > $OUT/1_base.log 2>&1; echo "[base] rc=$?"
timeout 300 python scripts/precond_apply_timing.py new > $OUT/2_new.log 2>&1; echo "[new] rc=$?"
cp gpurun_out/precond_apply_timing_*.json $OUT/
python - <<'PY'
import json
a = json.load(open("gpurun_out/precond_apply_timing_base.json")); b = json.load(open("gpurun_out/precond_apply_timing_new.json"))
for x, y in zip(a, b):
    print(x["n"], x["rank"], x["columns"], "base %.1f us  new %.1f us  (%.2fx)  dev %.1e / %.1e" % (x["us_per_apply"], y["us_per_apply"], x["us_per_apply"] / y["us_per_apply"], x["max_abs_dev_vs_float64"], y["max_abs_dev_vs_float64"]))
PY
timeout 400 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_love_vs_oracle.py -m gpu -q -x 2>&1 | tail -2
