#!/bin/bash
# round 6, GPU session 45 (last): the structured-operator and C5 end-to-end tests that build MultitaskMultivariateNormals, after its rewrite
set +e
OUT=gpurun_out/r6s45; mkdir -p $OUT
export TMPDIR=/tmp
timeout 110 python -m pytest tests/test_gpu_structured.py tests/test_gpu_parity_at_size.py -x -q -m gpu -k "multitask or c5 or kronecker" > $OUT/1_structured_c5.log 2>&1; echo "[structured / c5] rc=$?"; tail -4 $OUT/1_structured_c5.log | cut -c1-220
