#!/bin/bash
# round 5, GPU session 7: the whole -m gpu suite on the round's binary (with durations), smoke()
set +e
OUT=gpurun_out/r5s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $OUT/1_gpu_pytest_full.log 2>&1; echo "[full gpu suite] rc=$?"; tail -60 $OUT/1_gpu_pytest_full.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -2 $OUT/2_smoke.log
