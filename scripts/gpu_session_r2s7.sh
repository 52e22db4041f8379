#!/bin/bash
set +e
OUT=gpurun_out/r2s7; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_compose.py tests/test_gpu_hadamard.py tests/test_gpu_grad2.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -40
timeout 600 python scripts/precond_apply_bench.py > $OUT/precond_bench.log 2>&1; cat $OUT/precond_bench.log
