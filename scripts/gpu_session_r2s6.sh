#!/bin/bash
set +e
OUT=gpurun_out/r2s6; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_extra.py tests/test_gpu_bbmm.py tests/test_gpu_model.py tests/test_gpu_grad2.py tests/test_gpu_generic.py tests/test_gpu_multitask.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -40
timeout 600 python scripts/grad_timing.py r2s6 > $OUT/grad.log 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/grad_timing_r2s6.json")):
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if not k.endswith("tflops")})
PY
timeout 900 python scripts/posterior_profile.py r2s6 > $OUT/posterior.log 2>&1; echo "posterior rc=$?"; tail -1 $OUT/posterior.log | cut -c1-1500
