#!/bin/bash
set +e
OUT=gpurun_out/r2s2; mkdir -p $OUT
./scripts/micro/mfma_rates > $OUT/mfma_rates.txt 2>&1; cat $OUT/mfma_rates.txt
timeout 1200 python -m pytest tests/test_gpu_kv.py tests/test_gpu_reference_examples.py tests/test_gpu_parity_at_size.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -30
timeout 900 python scripts/kv_small_t.py r2s2 > $OUT/kv_small_t.log 2>&1; echo "small_t rc=$?"; cat $OUT/kv_small_t.log | cut -c1-400
