#!/bin/bash
# round 6, GPU session 12: far-pair tile culling -- tests, K*V timing on the road-like cloud, the split kernels' time at the headline shapes against round 6's
# record (the tile-list pointer must cost nothing when culling is off), road3d with and without the cutoff
set +e
OUT=gpurun_out/r6s12; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_far_cull.py tests/test_gpu_kv_split.py -x -q -m gpu > $OUT/0_tests.log 2>&1; echo "[far cull + split tests] rc=$?"; tail -12 $OUT/0_tests.log
timeout 300 python scripts/kv_split_time.py r6s12 > $OUT/1_kv_split_time.log 2>&1; echo "[kv split time] rc=$?"; cut -c1-260 $OUT/1_kv_split_time.log | tail -9
timeout 600 python scripts/far_cull_timing.py > $OUT/2_far_cull_timing.log 2>&1; echo "[far cull timing] rc=$?"; grep '"t": 11' $OUT/2_far_cull_timing.log | cut -c1-700
timeout 300 python bench.py --config road3d > $OUT/3_bench_road3d.json 2> $OUT/3_bench_road3d.err; echo "[bench road3d] rc=$?"; cut -c1-1200 $OUT/3_bench_road3d.json
timeout 300 python bench.py --config road3d --far-cutoff 1e-7 > $OUT/4_bench_road3d_far.json 2> $OUT/4_bench_road3d_far.err; echo "[bench road3d far 1e-7] rc=$?"; cut -c1-1200 $OUT/4_bench_road3d_far.json
cp gpurun_out/far_cull_timing.json gpurun_out/kv_split_r6s12.json $OUT/ 2>/dev/null
