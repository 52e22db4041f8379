#!/bin/bash
set +e
OUT=gpurun_out/r2s34; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kv_split.py tests/test_gpu_bbmm.py tests/test_gpu_model.py tests/test_gpu_batch.py tests/test_gpu_hadamard.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -20
timeout 300 python scripts/cg_graph_timing.py r2s34 2>&1 | grep -E "^\{|rror" | cut -c1-200
