#!/bin/bash
# round 5, GPU session 4: hazard stress, high dimensions again (25..28-dimension workspace fix), C3 at size (cheaper), workloads after the quick wins
set +e
OUT=gpurun_out/r5s4; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_hazard_stress.py -m gpu -q > $OUT/1_hazard.log 2>&1; echo "[hazard stress] rc=$?"; tail -6 $OUT/1_hazard.log
timeout 600 python -m pytest tests/test_gpu_highdim.py tests/test_gpu_kv.py tests/test_gpu_recenter.py -m gpu -q -x > $OUT/2_kv.log 2>&1; echo "[highdim + kv + recenter] rc=$?"; tail -6 $OUT/2_kv.log
timeout 900 python -m pytest tests/test_gpu_c3_at_size.py -m gpu -q > $OUT/3_c3.log 2>&1; echo "[c3 at size] rc=$?"; tail -6 $OUT/3_c3.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_road -o road -- python $R/scripts/workload_breakdown.py road3d plain > $R/$OUT/5_road_plain.log 2>&1); echo "[road plain rocprof] rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_protein -o protein -- python $R/scripts/workload_breakdown.py protein plain > $R/$OUT/6_protein_plain.log 2>&1); echo "[protein plain rocprof] rc=$?"
grep -h "seconds_per_iteration" -A8 $OUT/5_road_plain.log $OUT/6_protein_plain.log | tr -d '\n ' | cut -c1-600; echo
for f in $(find $OUT/prof_road $OUT/prof_protein -name "*kernel_stats*.csv"); do echo $f; head -9 $f | cut -c1-150; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 300 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_structured.py tests/test_gpu_batch.py -m gpu -q -x > $OUT/7_tests.log 2>&1; echo "[bbmm structured batch] rc=$?"; tail -4 $OUT/7_tests.log
timeout 200 python bench.py --config c2 --steps 2 --warmup 1 --other-steps 1 --skip-cpu-baseline > $OUT/8_bench_c2.json 2> $OUT/8_bench_c2.err; echo "[bench c2] rc=$?"; cut -c1-700 $OUT/8_bench_c2.json
