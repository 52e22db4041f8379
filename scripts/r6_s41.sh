#!/bin/bash
# round 6, GPU session 41: the reference's example-test cases on the device (dense and BBMM branches); the modules whose code the NaN-policy / fantasy changes touch
set +e
OUT=gpurun_out/r6s41; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_reference_examples.py -q -m gpu --durations=12 > $OUT/1_reference_examples.log 2>&1; echo "[reference examples] rc=$?"; tail -32 $OUT/1_reference_examples.log | cut -c1-260
timeout 200 python -m pytest tests/test_gpu_model.py tests/test_gpu_extra.py -x -q -m gpu -k "fantasy or nan or mask or posterior or predict" > $OUT/2_model_extra.log 2>&1; echo "[model / extra: fantasy, nan policy, posterior] rc=$?"; tail -3 $OUT/2_model_extra.log | cut -c1-260
