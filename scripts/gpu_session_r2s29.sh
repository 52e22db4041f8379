#!/bin/bash
set +e
OUT=gpurun_out/r2s29; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_extra.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -20
timeout 600 python scripts/cg_graph_timing.py r2s29 > $OUT/timing.log 2>&1; echo "rc=$?"; grep -E "^\{|Error|error|Warn" $OUT/timing.log | cut -c1-300
