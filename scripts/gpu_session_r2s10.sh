#!/bin/bash
set +e
OUT=gpurun_out/r2s10; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_compose.py tests/test_gpu_kv.py tests/test_gpu_model.py tests/test_gpu_bbmm.py "tests/test_gpu_parity_at_size.py::test_c4_single_gpu_share_end_to_end" "tests/test_gpu_parity_at_size.py::test_c5_multitask_end_to_end" "tests/test_gpu_parity_at_size.py::test_c3_end_to_end_preconditioned_mll" -m gpu -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  |s call" $OUT/pytest.log | head -40
cat gpurun_out/c4_share_end_to_end.json gpurun_out/c5_end_to_end.json
