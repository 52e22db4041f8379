#!/bin/bash
# round 6, GPU session 9: preconditioner ranks beyond 128 (build, posterior), bench lines on the round's binary (metric under rocprofv3, c3, c5, c2 at --steps 3, road3d)
set +e
OUT=gpurun_out/r6s9; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_bbmm.py -m gpu -q -k "precond or pivoted" > $OUT/0_tests.log 2>&1; echo "[precond tests] rc=$?"; tail -3 $OUT/0_tests.log
timeout 300 python scripts/precond_build_timing.py > $OUT/1_precond_build.log 2>&1; echo "[precond build] rc=$?"; grep "'n'" $OUT/1_precond_build.log | cut -c1-220
timeout 500 python scripts/mll_precond_timing.py posterior 500000 > $OUT/2_posterior_ranks.log 2>&1; echo "[posterior by rank] rc=$?"; grep "max_preconditioner_size" $OUT/2_posterior_ranks.log | cut -c1-400
(cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --other-steps 1 > $R/$OUT/3_bench_metric.json 2> $R/$OUT/3_bench_metric.err); echo "[bench metric under rocprofv3] rc=$?"; cut -c1-600 $OUT/3_bench_metric.json
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -6 $f | cut -c1-200; cp $f $OUT/bench_kernel_stats.csv; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/4_bench_c3.json 2> $OUT/4_bench_c3.err; echo "[bench c3] rc=$?"; cut -c1-500 $OUT/4_bench_c3.json
timeout 500 python bench.py --config c5 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/5_bench_c5.json 2> $OUT/5_bench_c5.err; echo "[bench c5] rc=$?"; cut -c1-500 $OUT/5_bench_c5.json
timeout 200 python bench.py --config c2 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/6_bench_c2.json 2> $OUT/6_bench_c2.err; echo "[bench c2] rc=$?"; cut -c1-500 $OUT/6_bench_c2.json
timeout 200 python bench.py --config road3d > $OUT/7_bench_road3d.json 2> $OUT/7_bench_road3d.err; echo "[bench road3d] rc=$?"; cut -c1-900 $OUT/7_bench_road3d.json
timeout 100 python bench.py --config protein > $OUT/8_bench_protein.json 2> $OUT/8_bench_protein.err; echo "[bench protein] rc=$?"; cut -c1-700 $OUT/8_bench_protein.json
cp gpurun_out/precond_build_timing.json gpurun_out/mll_precond_timing_n500000.json $OUT/ 2>/dev/null
