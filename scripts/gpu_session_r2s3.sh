#!/bin/bash
set +e
OUT=gpurun_out/r2s3; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_grad2.py tests/test_gpu_reference_examples.py tests/test_gpu_model.py tests/test_gpu_edge.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -40
timeout 900 python scripts/kv_small_t.py r2s3 > $OUT/kv_small_t.log 2>&1; echo "small_t rc=$?"; python - <<'PY'
import json
for r in json.load(open("gpurun_out/kv_small_t_r2s3.json")):
    print(r["kind"], r["d"], r["t"], " ".join(f"{k}={r[k+'_ms']:.1f}ms/{r[k+'_frac_fp32_mfma_peak']:.3f}/{r[k+'_rel_err']:.1e}" for k in ("new","g4","wide")))
PY
timeout 900 python scripts/grad_timing.py r2s3 > $OUT/grad_timing.log 2>&1; echo "grad rc=$?"; cat $OUT/grad_timing.log | cut -c1-900
