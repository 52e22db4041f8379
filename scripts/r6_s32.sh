#!/bin/bash
# round 6, GPU session 32: the driver's bench command on the round's final binary; C3 line; the reference's two workloads
set +e
OUT=gpurun_out/r6s32; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/1_bench_default.json 2> $OUT/1_bench_default.err; echo "[bench --steps 20 --warmup 5] rc=$?"; cut -c1-400 $OUT/1_bench_default.json
timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --other-steps 3 --skip-cpu-baseline --skip-extras > $OUT/2_bench_c3.json 2> $OUT/2_bench_c3.err; echo "[bench c3] rc=$?"; cut -c1-300 $OUT/2_bench_c3.json
timeout 100 python bench.py --config protein > $OUT/3_bench_protein.json 2> $OUT/3_bench_protein.err; echo "[bench protein] rc=$?"; cut -c1-200 $OUT/3_bench_protein.json
timeout 200 python bench.py --config road3d > $OUT/4_bench_road3d.json 2> $OUT/4_bench_road3d.err; echo "[bench road3d] rc=$?"; cut -c1-200 $OUT/4_bench_road3d.json
