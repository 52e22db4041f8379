#!/bin/bash
# round 6, GPU session 26: same-box A/B of the protein-shaped closure -- library before / after the preconditioner-projection changes x region streams off / on
set +e
OUT=gpurun_out/r6s26; mkdir -p $OUT
export TMPDIR=/tmp
BASE=$GRAFT_REPO_ROOT/gpytorch_amd/csrc/tune/libgpamd_base.so
for rep in 1 2 3; do
for cfg in base0 base1 new0 new1; do
  case $cfg in base*) export GPAMD_LIBRARY=$BASE;; *) unset GPAMD_LIBRARY;; esac
  GPAMD_REGION_STREAMS=${cfg: -1} timeout 200 python scripts/workload_breakdown.py protein plain 12 > $OUT/plain_${cfg}_$rep.log 2>&1
  cp gpurun_out/workload_breakdown_protein_plain.json $OUT/protein_plain_${cfg}_$rep.json
  python -c "
import json, statistics
d = json.load(open('$OUT/protein_plain_${cfg}_$rep.json')); v = [1e3 * x for x in d['stages'][0]['seconds_per_iteration']][3:]
print('$cfg rep $rep: median %.2f min %.2f ms' % (statistics.median(v), min(v)))"
done
done
