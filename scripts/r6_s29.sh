#!/bin/bash
# round 6, GPU session 29: region streams off / on for the wide-cloud solve test and the generic-path test (interleaved, same box)
set +e
OUT=gpurun_out/r6s29; mkdir -p $OUT
for rep in 1 2; do for rs in 0 1; do
  GPAMD_REGION_STREAMS=$rs timeout 300 python -m pytest "tests/test_gpu_recenter.py::test_cg_solve_and_mll_on_a_wide_cloud" "tests/test_gpu_recenter.py::test_posterior_with_love_on_a_wide_cloud" -m gpu -q --durations=5 2>&1 | grep -E "call|passed|failed" | sed "s/^/rs=$rs rep=$rep: /"
done; done
