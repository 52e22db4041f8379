#!/bin/bash
set +e
OUT=gpurun_out/r2s15; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kv_split.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -20
timeout 600 python scripts/kv_split_time.py r2s15 > $OUT/time.log 2>&1; echo "time rc=$?"; grep -E "^\{|Error|error" $OUT/time.log | head -20
