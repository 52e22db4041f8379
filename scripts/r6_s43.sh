#!/bin/bash
# round 6, GPU session 43: the at-size posterior tests (prediction strategy, LOVE, multitask) on the head after the host-layer sweep
set +e
OUT=gpurun_out/r6s43; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_dense_at_size.py tests/test_gpu_love_vs_oracle.py tests/test_gpu_recenter.py tests/test_gpu_c5_at_size.py -x -q -m gpu -k "posterior or love or c5" --durations=6 > $OUT/1_at_size_posteriors.log 2>&1; echo "[at-size posteriors] rc=$?"; tail -10 $OUT/1_at_size_posteriors.log | cut -c1-220
