#!/bin/bash
# round 6, GPU session 1: MLL forward + backward per preconditioner rank and the cold posterior per (rank, tolerance, LOVE rank) at the metric shape
set +e
OUT=gpurun_out/r6s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/mll_precond_timing.py ${1:-both} 500000 > $OUT/1_mll_precond.log 2>&1; echo "[mll precond] rc=$?"; tail -30 $OUT/1_mll_precond.log
cp gpurun_out/mll_precond_timing_n500000.json $OUT/ 2>/dev/null
