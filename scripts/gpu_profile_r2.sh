#!/bin/bash
# rocprofv3 evidence for round 2: kernel stats of the bench command + PMC passes (separate runs per counter group, as the
# microarchitecture guide prescribes) for the dominant t = 65 kernel and the small-t kernels (t = 1, 11, 16, 17).
# Usage: gpurun -- 'bash scripts/gpu_profile_r2.sh <tag>'
set +e
TAG=${1:-r2prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocprof stats"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-extras --skip-parity > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-200; done
pmc() { t=$1; name=$2; shift; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_t${t}_$name -o pmc -- python $R/scripts/kv_only.py 500000 $t 2 > $R/$OUT/pmc_t${t}_$name.log 2>&1); echo "pmc t=$t $name rc=$?"; }
for t in 65 1 11 16 17; do
  pmc $t mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
  pmc $t insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  pmc $t fetch FETCH_SIZE
  pmc $t write WRITE_SIZE
done
find $OUT -name "*kernel_trace*" -size +5M -delete
python scripts/collect_profiles_r2.py $TAG
