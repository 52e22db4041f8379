#!/bin/bash
# round 5, GPU session 3: 17..32 dimensions, C3 at size vs float64 BBMM, f-variance with refinement, generic path re-targeted, LDS ablations' clock / power
set +e
OUT=gpurun_out/r5s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_highdim.py -m gpu -q > $OUT/1_highdim.log 2>&1; echo "[highdim] rc=$?"; tail -12 $OUT/1_highdim.log
timeout 600 python scripts/kv_highdim_timing.py 500000 > $OUT/2_highdim_timing.log 2>&1; echo "[highdim timing] rc=$?"; tail -14 $OUT/2_highdim_timing.log | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_c3_at_size.py -m gpu -q > $OUT/3_c3.log 2>&1; echo "[c3 at size] rc=$?"; tail -12 $OUT/3_c3.log
timeout 900 python -m pytest tests/test_gpu_dense_at_size.py -m gpu -q -k posterior > $OUT/4_posterior.log 2>&1; echo "[posterior at size] rc=$?"; tail -8 $OUT/4_posterior.log
timeout 600 python -m pytest tests/test_gpu_generic.py tests/test_gpu_love_vs_oracle.py -m gpu -q > $OUT/5_generic_love.log 2>&1; echo "[generic + love] rc=$?"; tail -8 $OUT/5_generic_love.log
timeout 200 python scripts/kgh_clock_power.py r5s3 500000 3 > $OUT/6_clock_power.log 2>&1; echo "[clock power] rc=$?"; cut -c1-230 $OUT/6_clock_power.log
