#!/bin/bash
# round 6, GPU session 36: one- / two-column products with the generation written stage by stage (no dependent-instruction s_nops) -- same-box A/B
# against the previous library, the kv / model parity tests on the new one, the road3d-shaped prediction
set +e
OUT=gpurun_out/r6s36; mkdir -p $OUT
BASE=$GRAFT_REPO_ROOT/gpytorch_amd/csrc/tune/libgpamd_base.so
for rep in 1 2; do
  GPAMD_LIBRARY=$BASE timeout 300 python scripts/kv_few_cols_timing.py base$rep > $OUT/base$rep.log 2>&1; echo "[base $rep] rc=$?"
  timeout 300 python scripts/kv_few_cols_timing.py new$rep > $OUT/new$rep.log 2>&1; echo "[new $rep] rc=$?"
done
cp gpurun_out/kv_few_cols_timing_*.json $OUT/
python - <<'PY'
import json
a1, a2, b1, b2 = (json.load(open(f"gpurun_out/kv_few_cols_timing_{t}.json")) for t in ("base1", "base2", "new1", "new2"))
for w, x, y, z in zip(a1, a2, b1, b2):
    base, new = min(w["ms"], x["ms"]), min(y["ms"], z["ms"])
    print(w["kind"], w["n"], w["d"], w["t"], "base %.3f new %.3f ms (%.2fx)  dev %.1e / %.1e" % (base, new, base / new, w["rel_dev_vs_own_rows"], y["rel_dev_vs_own_rows"]))
PY
timeout 600 python -m pytest tests/test_gpu_kv.py tests/test_gpu_model.py tests/test_gpu_hazard_stress.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "[kv / model / hazard tests] rc=$?"; tail -3 $OUT/tests.log
timeout 200 python scripts/road3d_predict_profile.py > $OUT/road3d_predict.log 2>&1; echo "[road3d predict] rc=$?"; tail -2 $OUT/road3d_predict.log | cut -c1-400
