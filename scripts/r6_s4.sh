#!/bin/bash
# round 6, GPU session 4: one-launch-per-step pivoted Cholesky (parity + timing), variational predictive variance (at-size posterior tests, multitask),
# kv_gramh fence A/B, repaired "no contraction MFMAs" ablation with clock / power, protein-shaped closure
set +e
OUT=gpurun_out/r6s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_multitask.py tests/test_gpu_structured.py -m gpu -q -x > $OUT/1_tests.log 2>&1; echo "[bbmm / multitask / structured tests] rc=$?"; tail -6 $OUT/1_tests.log
timeout 300 python scripts/precond_build_timing.py > $OUT/2_precond_build.log 2>&1; echo "[precond build] rc=$?"; tail -12 $OUT/2_precond_build.log
timeout 400 python -m pytest tests/test_gpu_dense_at_size.py -m gpu -q -k posterior --durations=5 > $OUT/3_posterior_at_size.log 2>&1; echo "[posterior at size] rc=$?"; tail -12 $OUT/3_posterior_at_size.log
timeout 300 python scripts/kv_gramh_fence_ab.py $OUT/kv_gramh_fence_ab.json > $OUT/4_fence_ab.log 2>&1; echo "[gramh fence A/B] rc=$?"; tail -8 $OUT/4_fence_ab.log
timeout 300 python scripts/kgh_clock_power.py r6 500000 3 > $OUT/5_clock_power.log 2>&1; echo "[clock / power] rc=$?"; tail -12 $OUT/5_clock_power.log
timeout 300 python bench.py --config protein > $OUT/6_bench_protein.json 2> $OUT/6_bench_protein.err; echo "[protein] rc=$?"; cut -c1-1500 $OUT/6_bench_protein.json
cp gpurun_out/posterior_at_size_*.json gpurun_out/precond_build_timing.json gpurun_out/kgh_clock_power_r6.json $OUT/ 2>/dev/null
