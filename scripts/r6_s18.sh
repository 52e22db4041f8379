#!/bin/bash
# round 6, GPU session 18: the driver's own sequence on the round's final binary -- full -m gpu suite (durations), smoke(), the default bench command --, the
# bench command under rocprofv3 (kernel statistics), the C3 / C5 / C2 lines at --steps 3
set +e
OUT=gpurun_out/r6s18; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=40 > $OUT/1_gpu_suite.log 2>&1; echo "[gpu suite] rc=$?"; tail -4 $OUT/1_gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/2_smoke.log 2>&1; echo "[smoke] rc=$?"; tail -1 $OUT/2_smoke.log
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/3_bench_default.json 2> $OUT/3_bench_default.err; echo "[bench --steps 20 --warmup 5] rc=$?"; cut -c1-400 $OUT/3_bench_default.json
(cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --other-steps 1 --skip-extras > $R/$OUT/4_bench_metric_rocprof.json 2> $R/$OUT/4_bench_metric_rocprof.err); echo "[bench metric under rocprofv3] rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -6 $f | cut -c1-200; cp $f $OUT/bench_kernel_stats.csv; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/5_bench_c3.json 2> $OUT/5_bench_c3.err; echo "[bench c3] rc=$?"; cut -c1-300 $OUT/5_bench_c3.json
timeout 500 python bench.py --config c5 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/6_bench_c5.json 2> $OUT/6_bench_c5.err; echo "[bench c5] rc=$?"; cut -c1-300 $OUT/6_bench_c5.json
timeout 200 python bench.py --config c2 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/7_bench_c2.json 2> $OUT/7_bench_c2.err; echo "[bench c2] rc=$?"; cut -c1-300 $OUT/7_bench_c2.json
timeout 100 python bench.py --config protein > $OUT/8_bench_protein.json 2> $OUT/8_bench_protein.err; echo "[bench protein] rc=$?"; cut -c1-500 $OUT/8_bench_protein.json
