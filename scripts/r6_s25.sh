#!/bin/bash
# round 6, GPU session 25: 32-row basis tiles of the preconditioner's projection kernel at small n -- tests, protein-shaped closure, kernel statistics
set +e
OUT=gpurun_out/r6s25; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_love_vs_oracle.py tests/test_gpu_model.py -m gpu -q -x > $OUT/1_tests.log 2>&1; echo "[bbmm / love / model tests] rc=$?"; tail -4 $OUT/1_tests.log
timeout 200 python scripts/workload_breakdown.py protein plain 10 > $OUT/2_protein_plain.log 2>&1; echo "[protein plain] rc=$?"
cp gpurun_out/workload_breakdown_protein_plain.json $OUT/protein_plain.json
python -c "import json; d = json.load(open('$OUT/protein_plain.json')); print([round(1e3 * x, 2) for x in d['stages'][0]['seconds_per_iteration']])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_protein -o prof -- python $GRAFT_REPO_ROOT/scripts/workload_breakdown.py protein plain 6 > $GRAFT_REPO_ROOT/$OUT/3_protein_rocprof.log 2>&1); echo "[protein plain under rocprof] rc=$?"
find $OUT/prof_protein -name "*kernel_stats.csv" -exec cp {} $OUT/protein_kernel_stats.csv \;
rm -rf $OUT/prof_protein
head -30 $OUT/protein_kernel_stats.csv | cut -c1-150
