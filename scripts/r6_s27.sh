#!/bin/bash
# round 6, GPU session 27: same-box A/B after the sum / apply changes; road3d-shaped iteration with region streams off / on, interleaved; tests; kernel statistics
set +e
OUT=gpurun_out/r6s27; mkdir -p $OUT
export TMPDIR=/tmp
BASE=$GRAFT_REPO_ROOT/gpytorch_amd/csrc/tune/libgpamd_base.so
timeout 500 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_love_vs_oracle.py tests/test_gpu_model.py -m gpu -q -x > $OUT/1_tests.log 2>&1; echo "[bbmm / love / model tests] rc=$?"; tail -2 $OUT/1_tests.log
for rep in 1 2 3; do
for cfg in base0 new0 new1; do
  case $cfg in base*) export GPAMD_LIBRARY=$BASE;; *) unset GPAMD_LIBRARY;; esac
  GPAMD_REGION_STREAMS=${cfg: -1} timeout 200 python scripts/workload_breakdown.py protein plain 12 > $OUT/plain_${cfg}_$rep.log 2>&1
  cp gpurun_out/workload_breakdown_protein_plain.json $OUT/protein_plain_${cfg}_$rep.json
  python -c "
import json, statistics
d = json.load(open('$OUT/protein_plain_${cfg}_$rep.json')); v = [1e3 * x for x in d['stages'][0]['seconds_per_iteration']][3:]
print('$cfg rep $rep: median %.2f min %.2f ms' % (statistics.median(v), min(v)))"
done
done
unset GPAMD_LIBRARY
for rep in 1 2; do
for rs in 0 1; do
  GPAMD_REGION_STREAMS=$rs timeout 300 python bench.py --config road3d --steps 10 > $OUT/road3d_rs${rs}_$rep.json 2> $OUT/road3d_rs${rs}_$rep.err
  python -c "
import json
d = json.load(open('$OUT/road3d_rs${rs}_$rep.json')); print('road3d region_streams=$rs rep $rep: median %.4f s' % d['value'], [round(x, 3) for x in d['config']['seconds_per_iteration']])"
done
done
(cd /tmp && GPAMD_REGION_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_protein -o prof -- python $GRAFT_REPO_ROOT/scripts/workload_breakdown.py protein plain 6 > $GRAFT_REPO_ROOT/$OUT/3_protein_rocprof.log 2>&1); echo "[protein plain under rocprof] rc=$?"
find $OUT/prof_protein -name "*kernel_stats.csv" -exec cp {} $OUT/protein_kernel_stats.csv \;
rm -rf $OUT/prof_protein
head -14 $OUT/protein_kernel_stats.csv | cut -c1-150
