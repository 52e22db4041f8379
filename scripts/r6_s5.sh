#!/bin/bash
# round 6, GPU session 5: variational variance vs solve tolerance; kv_gramh with 8 tied wait states (A/B against round 5's form and the full fence);
# hazard stress + kv tests on the new binary; bench strong test; new oracle-coincidence tests
set +e
OUT=gpurun_out/r6s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python scripts/variational_variance_sweep.py > $OUT/1_variational_sweep.log 2>&1; echo "[variational sweep] rc=$?"; grep "precond_rank" $OUT/1_variational_sweep.log
timeout 300 python scripts/kv_gramh_fence_ab.py $OUT/kv_gramh_fence_ab.json > $OUT/2_fence_ab.log 2>&1; echo "[gramh fence A/B] rc=$?"; tail -6 $OUT/2_fence_ab.log | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_hazard_stress.py tests/test_gpu_bench_strong.py tests/test_gpu_love_vs_oracle.py -m gpu -q > $OUT/3_tests.log 2>&1; echo "[hazard / bench strong / love-vs-oracle] rc=$?"; tail -25 $OUT/3_tests.log
timeout 300 python -m pytest tests/test_gpu_grad_at_size.py -m gpu -q -k "bilinear" > $OUT/4_grad_blocks.log 2>&1; echo "[grad blocks] rc=$?"; tail -12 $OUT/4_grad_blocks.log
cp gpurun_out/posterior_at_size_c2_variational.json gpurun_out/default_settings_vs_oracle.json gpurun_out/love_vs_oracle.json $OUT/ 2>/dev/null
