#!/bin/bash
# round 6, GPU session 38 (last): the hazard stress tests on the REBUILT tune library (its objects were older than common.hpp's staged pair
# generation during session 37), then the bench command of the head under rocprofv3 (kernel statistics of the final binary)
set +e
OUT=gpurun_out/r6s38; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_hazard_stress.py -x -q -m gpu > $OUT/1_hazard_stress.log 2>&1; echo "[hazard stress on the rebuilt tune library] rc=$?"; tail -2 $OUT/1_hazard_stress.log
(cd /tmp && timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --other-steps 1 --skip-extras --skip-cpu-baseline > $R/$OUT/2_bench_metric_rocprof.json 2> $R/$OUT/2_bench_metric_rocprof.err); echo "[bench metric under rocprofv3] rc=$?"
cut -c1-300 $OUT/2_bench_metric_rocprof.json
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-220; cp $f $OUT/bench_kernel_stats.csv; done
find $OUT -name "*kernel_trace*" -delete
find $OUT/prof -type f -size +3M -delete
