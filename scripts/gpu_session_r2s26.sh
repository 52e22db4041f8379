#!/bin/bash
set +e
OUT=gpurun_out/r2s26; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kv_split.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -20
timeout 300 python scripts/kgh_ablate.py r2s26 500000 0:2,0:2 2>&1 | grep -E "^\{|rror" | cut -c1-200
timeout 600 python scripts/kv_split_time.py r2s26 > $OUT/time.log 2>&1; echo "rc=$?"; grep -E "^\{|Error|error" $OUT/time.log | cut -c1-420
