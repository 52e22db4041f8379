#!/bin/bash
# end-of-round verification: whole -m gpu suite on the library defaults + smoke, the core files again on the fp32-MFMA contraction, the bench line
set +e
OUT=gpurun_out/r2s31; mkdir -p $OUT
bash scripts/gpu_tests.sh r2s31
GPAMD_KV_SPLIT=0 timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_bbmm.py tests/test_gpu_model.py tests/test_gpu_multitask.py tests/test_gpu_reference_examples.py -m gpu -q -p no:cacheprovider > $OUT/pytest_f32.log 2>&1; echo "pytest (GPAMD_KV_SPLIT=0) rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_f32.log | tail -5
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
