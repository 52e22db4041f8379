#!/bin/bash
# round 6, GPU session 39: the product on the device against the reference's PUBLISHED notebook outputs (dense and BBMM branches); the tests that
# touch FixedNoiseGaussianLikelihood after its learned noise took the likelihood's batch_shape
set +e
OUT=gpurun_out/r6s39; mkdir -p $OUT
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_published_runs.py -q -m gpu --durations=10 > $OUT/1_published_runs.log 2>&1; echo "[published runs] rc=$?"; tail -25 $OUT/1_published_runs.log | cut -c1-400
timeout 200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_reference_examples.py -x -q -m gpu -k "fixed or Fixed or noise or batch" > $OUT/2_fixed_noise_tests.log 2>&1; echo "[fixed-noise / batch tests] rc=$?"; tail -3 $OUT/2_fixed_noise_tests.log
