#!/bin/bash
# round 6, GPU session 8: re-run of the tests changed since session 7
set +e
OUT=gpurun_out/r6s8; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_love_vs_oracle.py tests/test_gpu_recenter.py tests/test_gpu_dense_at_size.py tests/test_gpu_multitask.py tests/test_gpu_extra.py tests/test_gpu_structured.py -m gpu -q -k "love or default or extent or posterior or refinement or root or fantasy or sharded or split" > $OUT/1_tests.log 2>&1; echo "[tests] rc=$?"; tail -30 $OUT/1_tests.log
cp gpurun_out/default_settings_vs_oracle.json gpurun_out/love_vs_oracle_c3_model.json gpurun_out/posterior_at_size_*.json $OUT/ 2>/dev/null
