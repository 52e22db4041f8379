#!/bin/bash
set +e
OUT=gpurun_out/r2s5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_model.py tests/test_gpu_grad2.py tests/test_gpu_extra.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -30
GPAMD_GRAD2_MAXCOLS=66 timeout 600 python scripts/grad_timing.py r2s5_66 > $OUT/grad66.log 2>&1
timeout 600 python scripts/grad_timing.py r2s5_34 > $OUT/grad34.log 2>&1
GPAMD_GRAD2_MAXCOLS=18 timeout 600 python scripts/grad_timing.py r2s5_18 > $OUT/grad18.log 2>&1
python - <<'PY'
import json
for tag in ("66", "34", "18"):
    for r in json.load(open(f"gpurun_out/grad_timing_r2s5_{tag}.json")):
        print(tag, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if not k.endswith("tflops")})
PY
timeout 900 python scripts/posterior_profile.py r2s5 > $OUT/posterior.log 2>&1; echo "posterior rc=$?"; tail -8 $OUT/posterior.log | cut -c1-1200
