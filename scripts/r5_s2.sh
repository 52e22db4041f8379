#!/bin/bash
# round 5, GPU session 2: block Lanczos on the native kernels (test + timings at C2 and n = 500 000), sustained f16 MFMA clock / power
set +e
OUT=gpurun_out/r5s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_love_vs_oracle.py -m gpu -q -x > $OUT/1_love_test.log 2>&1; echo "[love test] rc=$?"; tail -6 $OUT/1_love_test.log
timeout 600 python scripts/love_block_timing.py c2 > $OUT/2_love_c2.log 2>&1; echo "[love c2] rc=$?"; tail -10 $OUT/2_love_c2.log
timeout 900 python scripts/love_block_timing.py metric > $OUT/3_love_metric.log 2>&1; echo "[love metric] rc=$?"; tail -10 $OUT/3_love_metric.log
for w in 1 2 4; do
  (./scripts/micro/mfma_f16_sustained 5 $w > $OUT/4_mfma_sustained_w$w.log 2>&1 &)
  sleep 2.0
  for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $OUT/4_mfma_sustained_w$w.smi; sleep 0.4; done
  sleep 2.5
  echo "[mfma sustained w=$w]"; tail -2 $OUT/4_mfma_sustained_w$w.log; tail -2 $OUT/4_mfma_sustained_w$w.smi
done
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_reference_examples.py tests/test_gpu_structured.py tests/test_gpu_extra.py -m gpu -q -x > $OUT/5_tests.log 2>&1; echo "[tests] rc=$?"; tail -5 $OUT/5_tests.log
