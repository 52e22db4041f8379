#!/bin/bash
# round 6, GPU session 15: far-cull tests on the final form; where a culled road3d iteration spends its time (rocprofv3 kernel stats of 4 iterations from lengthscale 0.05)
set +e
OUT=gpurun_out/r6s15; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_far_cull.py -q -m gpu > $OUT/0_tests.log 2>&1; echo "[far cull tests] rc=$?"; tail -4 $OUT/0_tests.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o road -- python $R/bench.py --config road3d --far-cutoff 1e-7 --steps 4 > $R/$OUT/1_road3d_far.json 2> $R/$OUT/1_road3d_far.err); echo "[road3d far under rocprofv3] rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f | cut -c1-200; cp $f $OUT/road3d_far_kernel_stats.csv; done
find $OUT -name "*kernel_trace*" -size +5M -delete
