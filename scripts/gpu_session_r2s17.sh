#!/bin/bash
set +e
OUT=gpurun_out/r2s17; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kv_split.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -20
for ni in 2 4; do
  GPAMD_KGH_NI=$ni timeout 300 python scripts/kv_split_time.py r2s17_ni$ni 500000 64,65,32,11 > $OUT/time_ni$ni.log 2>&1; echo "ni=$ni rc=$?"; grep -E "^\{|Error|error" $OUT/time_ni$ni.log | cut -c1-420
done
