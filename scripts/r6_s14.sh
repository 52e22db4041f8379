#!/bin/bash
# round 6, GPU session 14: kv_directh with two column tiles + extra column; culling tests; the love-vs-oracle module with the oracle's dense operator on the device
set +e
OUT=gpurun_out/r6s14; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kv_split.py tests/test_gpu_far_cull.py tests/test_gpu_kv.py tests/test_gpu_recenter.py -q -m gpu -x > $OUT/0_tests.log 2>&1; echo "[kv tests] rc=$?"; tail -8 $OUT/0_tests.log
timeout 400 python -m pytest tests/test_gpu_love_vs_oracle.py -q -m gpu > $OUT/1_love.log 2>&1; echo "[love vs oracle] rc=$?"; tail -12 $OUT/1_love.log
timeout 300 python scripts/kv_direct_split_timing.py > $OUT/2_direct_split_timing.log 2>&1; echo "[direct split timing] rc=$?"; tail -14 $OUT/2_direct_split_timing.log | cut -c1-300
timeout 600 python scripts/far_cull_timing.py > $OUT/3_far_cull_timing.log 2>&1; echo "[far cull timing] rc=$?"; grep -E '"t": (33|65)' $OUT/3_far_cull_timing.log | grep matern52 | cut -c1-420
cp gpurun_out/far_cull_timing.json gpurun_out/kv_direct_split_timing.json $OUT/ 2>/dev/null
