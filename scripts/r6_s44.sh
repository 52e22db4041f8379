#!/bin/bash
# round 6, GPU session 44 (last): the multitask modules on the device after the MultitaskMultivariateNormal rewrite
set +e
OUT=gpurun_out/r6s44; mkdir -p $OUT
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_multitask.py tests/test_gpu_reference_examples.py tests/test_gpu_c5_at_size.py tests/test_gpu_hadamard.py -x -q -m gpu -k "multitask or c5 or kronecker or hadamard" > $OUT/1_multitask_modules.log 2>&1; echo "[multitask modules] rc=$?"; tail -4 $OUT/1_multitask_modules.log | cut -c1-220
