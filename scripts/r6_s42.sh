#!/bin/bash
# round 6, GPU session 42: every model-level module after the constraint / distribution / prediction-strategy changes of the host-layer sweep
set +e
OUT=gpurun_out/r6s42; mkdir -p $OUT
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_model.py tests/test_gpu_bbmm.py tests/test_gpu_batch.py tests/test_gpu_compose.py tests/test_gpu_hadamard.py tests/test_gpu_multitask.py tests/test_gpu_reference_examples.py tests/test_gpu_published_runs.py tests/test_gpu_extra.py tests/test_gpu_structured.py tests/test_gpu_generic.py tests/test_gpu_edge.py -x -q -m gpu --durations=8 > $OUT/1_model_level_modules.log 2>&1; echo "[model-level modules] rc=$?"; tail -14 $OUT/1_model_level_modules.log | cut -c1-220
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -1 $OUT/2_smoke.log
