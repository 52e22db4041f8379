#!/bin/bash
# round 6, GPU session 24: region launches on a second stream -- bitwise test, protein-shaped closure A/B, road3d-shaped iteration A/B, kernel statistics
set +e
OUT=gpurun_out/r6s24; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_recenter.py -m gpu -q -x > $OUT/1_tests.log 2>&1; echo "[recenter tests] rc=$?"; tail -5 $OUT/1_tests.log
for rs in 0 1; do
  GPAMD_REGION_STREAMS=$rs timeout 200 python scripts/workload_breakdown.py protein plain 10 > $OUT/2_protein_plain_rs$rs.log 2>&1; echo "[protein plain region_streams=$rs] rc=$?"
  cp gpurun_out/workload_breakdown_protein_plain.json $OUT/protein_plain_rs$rs.json
  python -c "import json; d = json.load(open('$OUT/protein_plain_rs$rs.json')); print([round(1e3 * x, 2) for x in d['stages'][0]['seconds_per_iteration']])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_protein -o prof -- python $GRAFT_REPO_ROOT/scripts/workload_breakdown.py protein plain 6 > $GRAFT_REPO_ROOT/$OUT/3_protein_rocprof.log 2>&1); echo "[protein plain under rocprof] rc=$?"
find $OUT/prof_protein -name "*kernel_stats.csv" -exec cp {} $OUT/protein_kernel_stats.csv \;
rm -rf $OUT/prof_protein
head -12 $OUT/protein_kernel_stats.csv | cut -c1-160
for rs in 0 1; do
  GPAMD_REGION_STREAMS=$rs timeout 300 python bench.py --config road3d --steps 12 > $OUT/4_road3d_rs$rs.json 2> $OUT/4_road3d_rs$rs.err; echo "[road3d region_streams=$rs] rc=$?"; cut -c1-900 $OUT/4_road3d_rs$rs.json
done
