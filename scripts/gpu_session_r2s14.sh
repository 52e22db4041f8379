#!/bin/bash
set +e
OUT=gpurun_out/r2s14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_grad2.py tests/test_gpu_compose.py tests/test_gpu_hadamard.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -12
timeout 300 python scripts/grad_timing.py r2s14 > $OUT/grad.log 2>&1; python - <<PY
import json
for r in json.load(open("gpurun_out/grad_timing_r2s14.json")):
    print("grad2", r["kind"], {k: round(v, 1) for k, v in r.items() if (k.startswith("grad2") or k.startswith("kv")) and k.endswith("_ms")}, r["max_rel_dev_vs_direct"])
PY
