#!/bin/bash
# round 6, GPU session 47 (the last seconds): the prior-using device tests after the prior closures became classes
set +e
OUT=gpurun_out/r6s47; mkdir -p $OUT
timeout 40 python -m pytest tests/test_gpu_batch.py tests/test_gpu_reference_examples.py -x -q -m gpu -k "train_on_batch or case_prior or case_posterior_with_optimization or shared_hypers" > $OUT/1_prior_tests.log 2>&1; echo "[prior-using tests] rc=$?"; tail -2 $OUT/1_prior_tests.log | cut -c1-200
