#!/bin/bash
set +e
OUT=gpurun_out/r2s11; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_compose.py tests/test_gpu_kv.py tests/test_gpu_model.py tests/test_gpu_grad2.py "tests/test_gpu_parity_at_size.py::test_c4_single_gpu_share_end_to_end" "tests/test_gpu_parity_at_size.py::test_c5_multitask_end_to_end" -m gpu -q -p no:cacheprovider --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  |s call" $OUT/pytest.log | head -30
cat gpurun_out/c4_share_end_to_end.json gpurun_out/c5_end_to_end.json
for sk in 0 2 4 8 16; do GPAMD_KV_SKEW=$sk timeout 300 python scripts/kv_time.py 500000 65 4 2>/dev/null | tail -1; done | tee $OUT/kv_skew.jsonl
for sk in 0 3 5 8 12; do GPAMD_GRAD2_SKEW=$sk timeout 300 python scripts/grad_timing.py r2s11_sk$sk > $OUT/grad_sk$sk.log 2>&1; python - <<PY
import json
r = json.load(open("gpurun_out/grad_timing_r2s11_sk$sk.json"))[0]
print("grad2 skew $sk", {k: round(v, 1) for k, v in r.items() if k.startswith("grad2") and k.endswith("_ms")})
PY
done
