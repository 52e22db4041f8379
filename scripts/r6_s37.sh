#!/bin/bash
# round 6, GPU session 37 (final): the full -m gpu suite + smoke on the head (region streams, projection-kernel changes, two-column pair form)
set +e
OUT=gpurun_out/r6s37; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1300 python -m pytest tests/ -x -q -m gpu --durations=90 > $OUT/1_gpu_suite.log 2>&1; echo "[gpu suite] rc=$?"; tail -4 $OUT/1_gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -2 $OUT/2_smoke.log
