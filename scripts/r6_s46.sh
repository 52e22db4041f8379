#!/bin/bash
# round 6, GPU session 46 (the round's last seconds): the model-API tests of the modules not re-run since the host-layer sweep
set +e
OUT=gpurun_out/r6s46; mkdir -p $OUT
export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_gpu_far_cull.py tests/test_gpu_highdim.py tests/test_gpu_grad2.py tests/test_gpu_recenter.py -x -q -m gpu -k "model_api or beyond_16 or posterior_at_20 or harness_gradient or input_gradients or train_inputs or cg_solve_and_mll" > $OUT/1_model_api_rest.log 2>&1; echo "[model-API tests of the remaining modules] rc=$?"; tail -4 $OUT/1_model_api_rest.log | cut -c1-220
