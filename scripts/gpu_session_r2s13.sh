#!/bin/bash
set +e
OUT=gpurun_out/r2s13; mkdir -p $OUT
for pp in 1 0; do
  GPAMD_GRAD2_PIPE=$pp timeout 900 python -m pytest tests/test_gpu_grad2.py tests/test_gpu_compose.py tests/test_gpu_hadamard.py -m gpu -q -p no:cacheprovider > $OUT/pytest_pipe$pp.log 2>&1; echo "pytest pipe=$pp rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_pipe$pp.log | head -12
  GPAMD_GRAD2_PIPE=$pp timeout 300 python scripts/grad_timing.py r2s13_pipe$pp > $OUT/grad_pipe$pp.log 2>&1; python - <<PY
import json
for r in json.load(open("gpurun_out/grad_timing_r2s13_pipe$pp.json")):
    print("grad2 pipe $pp", r["kind"], {k: round(v, 1) for k, v in r.items() if (k.startswith("grad2") or k.startswith("kv")) and k.endswith("_ms")}, r["max_rel_dev_vs_direct"])
PY
done
