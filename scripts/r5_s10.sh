#!/bin/bash
# round 5, GPU session 10: the whole -m gpu suite on the round's FINAL binary and tests (for the record), smoke()
set +e
OUT=gpurun_out/r5s10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/1_gpu_pytest_full.log 2>&1; echo "[full gpu suite] rc=$?"; tail -25 $OUT/1_gpu_pytest_full.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -2 $OUT/2_smoke.log
