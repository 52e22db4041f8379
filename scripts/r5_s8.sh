#!/bin/bash
# round 5, GPU session 8: C3-at-size in its final form; bench lines on the round's binary (metric under rocprofv3, c3, c5, c2, road3d, protein)
set +e
OUT=gpurun_out/r5s8; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_c3_at_size.py -m gpu -q > $OUT/1_c3_at_size.log 2>&1; echo "[c3 at size] rc=$?"; tail -3 $OUT/1_c3_at_size.log; cp gpurun_out/c3_at_size_vs_float64.json $OUT/1_c3_at_size_vs_float64.json
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --other-steps 1 > $R/$OUT/2_bench_metric.json 2> $R/$OUT/2_bench_metric.err); echo "[bench metric under rocprofv3] rc=$?"; cut -c1-900 $OUT/2_bench_metric.json
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -6 $f | cut -c1-200; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 300 python bench.py --config c3 --steps 3 --warmup 1 --other-steps 2 --skip-cpu-baseline --skip-extras > $OUT/3_bench_c3.json 2> $OUT/3_bench_c3.err; echo "[bench c3] rc=$?"; cut -c1-700 $OUT/3_bench_c3.json
timeout 300 python bench.py --config c5 --steps 1 --warmup 1 --other-steps 1 --skip-cpu-baseline --skip-extras > $OUT/4_bench_c5.json 2> $OUT/4_bench_c5.err; echo "[bench c5] rc=$?"; cut -c1-700 $OUT/4_bench_c5.json
timeout 200 python bench.py --config c2 --steps 2 --warmup 1 --other-steps 2 --skip-cpu-baseline > $OUT/5_bench_c2.json 2> $OUT/5_bench_c2.err; echo "[bench c2] rc=$?"; cut -c1-700 $OUT/5_bench_c2.json
timeout 200 python bench.py --config road3d > $OUT/6_bench_road3d.json 2> $OUT/6_bench_road3d.err; echo "[bench road3d] rc=$?"; cut -c1-1200 $OUT/6_bench_road3d.json
timeout 100 python bench.py --config protein > $OUT/7_bench_protein.json 2> $OUT/7_bench_protein.err; echo "[bench protein] rc=$?"; cut -c1-1200 $OUT/7_bench_protein.json
