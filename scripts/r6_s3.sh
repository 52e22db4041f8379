#!/bin/bash
# round 6, GPU session 3: the new strong-scaling bench test, then the whole -m gpu suite with durations (planning the < 700 s budget)
set +e
OUT=gpurun_out/r6s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_strong.py -m gpu -q -x > $OUT/1_bench_strong.log 2>&1; echo "[bench strong] rc=$?"; tail -15 $OUT/1_bench_strong.log
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > $OUT/2_gpu_suite.log 2>&1; echo "[gpu suite] rc=$?"; tail -80 $OUT/2_gpu_suite.log
