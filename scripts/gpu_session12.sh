#!/bin/bash
set +e
OUT=gpurun_out/s12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_extra.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | tail
timeout 600 python scripts/kv_tune.py 100000 5 0,14,23,26,27 > $OUT/tune.log 2>&1; tail -7 $OUT/tune.log
