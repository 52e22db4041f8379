#!/bin/bash
# round 6, GPU session 11: the driver's own sequence on the final binary -- full -m gpu suite, smoke(), the default bench command -- plus the quadrature-steps sweep
set +e
OUT=gpurun_out/r6s11; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python scripts/mll_quadrature_steps.py > $OUT/0_quadrature_steps.log 2>&1; echo "[quadrature steps] rc=$?"; grep "quadrature_steps" $OUT/0_quadrature_steps.log | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/1_gpu_suite.log 2>&1; echo "[gpu suite] rc=$?"; tail -25 $OUT/1_gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -2 $OUT/2_smoke.log
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/3_bench_default.json 2> $OUT/3_bench_default.err; echo "[bench --steps 20 --warmup 5] rc=$?"; cut -c1-700 $OUT/3_bench_default.json
cp gpurun_out/mll_quadrature_steps.json $OUT/ 2>/dev/null
