#!/bin/bash
# rocprofv3 evidence of one round: kernel stats of the bench command + PMC passes (separate runs per counter group, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) for the dominant 65-column kernel on BOTH contraction paths and the small-t kernels.
# Usage: gpurun -- 'bash scripts/gpu_profile.sh <tag> [round]'   ->  gpurun_out/<tag>/..., summaries in profiles/r<round>_<tag>_*
set +e
TAG=${1:-prof}; ROUND=${2:-05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocprof stats"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --other-steps 1 --skip-cpu-baseline --skip-extras --skip-parity > $R/$OUT/rocprof_bench.json 2> $R/$OUT/rocprof.log); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-200; done
pmc() { path=$1; t=$2; name=$3; shift; shift; shift; (cd /tmp && GPAMD_KV_SPLIT=$([ $path = f32 ] && echo 0 || echo 1) timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${path}_t${t}_$name -o pmc -- python $R/scripts/kv_only.py 500000 $t 2 > $R/$OUT/pmc_${path}_t${t}_$name.log 2>&1); echo "pmc $path t=$t $name rc=$?"; }
for cfg in f32:65 split:65 split:1 split:11; do
  path=${cfg%%:*}; t=${cfg##*:}
  pmc $path $t mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
  pmc $path $t insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  pmc $path $t lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pmc $path $t fetch FETCH_SIZE
  pmc $path $t write WRITE_SIZE
done
find $OUT -name "*kernel_trace*" -size +5M -delete
python scripts/collect_profiles.py $TAG $ROUND
