#!/bin/bash
# round 6, GPU session 40: device twins of the host-layer sweep (composed-kernel known answers, kernel battery over seven families, RBF / periodic unit
# tests), the published-runs tests with the Dirichlet likelihood class, and every module whose kernels / likelihood code paths the sweep touched
set +e
OUT=gpurun_out/r6s40; mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/published_runs_on_device.json
timeout 300 python -m pytest tests/test_gpu_compose.py tests/test_gpu_published_runs.py -q -m gpu --durations=8 > $OUT/1_compose_published.log 2>&1; echo "[compose + published runs] rc=$?"; tail -30 $OUT/1_compose_published.log | cut -c1-300
cp gpurun_out/published_runs_on_device.json $OUT/ 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_reference_examples.py tests/test_gpu_model.py tests/test_gpu_hadamard.py tests/test_gpu_multitask.py tests/test_gpu_generic.py -x -q -m gpu > $OUT/2_touched_modules.log 2>&1; echo "[touched modules] rc=$?"; tail -4 $OUT/2_touched_modules.log | cut -c1-300
