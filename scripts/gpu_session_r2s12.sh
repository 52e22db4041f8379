#!/bin/bash
set +e
OUT=gpurun_out/r2s12; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_model.py "tests/test_gpu_parity_at_size.py::test_c4_single_gpu_share_end_to_end" "tests/test_gpu_parity_at_size.py::test_c5_multitask_end_to_end" -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -30
cat gpurun_out/c4_share_end_to_end.json gpurun_out/c5_end_to_end.json
for sk in 0 3 5 8 12; do GPAMD_GRAD2_SKEW=$sk timeout 300 python scripts/grad_timing.py r2s12_sk$sk > $OUT/grad_sk$sk.log 2>&1; python - <<PY
import json
for r in json.load(open("gpurun_out/grad_timing_r2s12_sk$sk.json")):
    print("grad2 skew $sk", r["kind"], {k: round(v, 1) for k, v in r.items() if k.startswith("grad2") and k.endswith("_ms")})
PY
done
