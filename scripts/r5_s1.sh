#!/bin/bash
# round 5, GPU session 1: multi-vector / block Lanczos on the device, LOVE timing at C2, workload breakdowns, clock / power of the split kernel
set +e
OUT=gpurun_out/r5s1; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_love_vs_oracle.py -m gpu -q -x > $OUT/1_love_test.log 2>&1; echo "[love test] rc=$?"; tail -15 $OUT/1_love_test.log
timeout 600 python scripts/love_block_timing.py c2 > $OUT/2_love_c2.log 2>&1; echo "[love c2] rc=$?"; tail -12 $OUT/2_love_c2.log
timeout 300 python scripts/workload_breakdown.py road3d phases > $OUT/3_road_phases.log 2>&1; echo "[road phases] rc=$?"
timeout 300 python scripts/workload_breakdown.py protein phases > $OUT/4_protein_phases.log 2>&1; echo "[protein phases] rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_road -o road -- python $R/scripts/workload_breakdown.py road3d plain > $R/$OUT/5_road_plain.log 2>&1); echo "[road plain rocprof] rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_protein -o protein -- python $R/scripts/workload_breakdown.py protein plain > $R/$OUT/6_protein_plain.log 2>&1); echo "[protein plain rocprof] rc=$?"
for f in $(find $OUT/prof_road $OUT/prof_protein -name "*kernel_stats*.csv"); do echo $f; head -12 $f | cut -c1-160; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 200 python scripts/kgh_clock_power.py r5s1 500000 4 > $OUT/7_clock_power.log 2>&1; echo "[clock power] rc=$?"; cat $OUT/7_clock_power.log | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_reference_examples.py -m gpu -q -x > $OUT/8_model_tests.log 2>&1; echo "[model tests] rc=$?"; tail -5 $OUT/8_model_tests.log
