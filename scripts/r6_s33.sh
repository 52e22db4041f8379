#!/bin/bash
# round 6, GPU session 33: why the accurate cold posterior of the bench extras went from 1.06 s to 4.35 s
set +e
OUT=gpurun_out/r6s33; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/posterior_accurate_profile.py > $OUT/1_plain.log 2>&1; echo "[plain] rc=$?"; grep rep $OUT/1_plain.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/scripts/posterior_accurate_profile.py > $GRAFT_REPO_ROOT/$OUT/2_rocprof.log 2>&1); echo "[rocprof] rc=$?"; grep rep $OUT/2_rocprof.log
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
head -16 $OUT/kernel_stats.csv | cut -c1-170
