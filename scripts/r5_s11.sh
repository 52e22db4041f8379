#!/bin/bash
# round 5, GPU session 11: PMC passes (separate runs per counter group) of the two 65-column kernels on the round's final binary -> profiles/kv_pmc_current.json,
# kv_pmc_split_current.json (what bench.py reports as roofline.traffic)
set +e
TAG=r5s11; ROUND=05
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pmc() { path=$1; t=$2; name=$3; shift; shift; shift; (cd /tmp && GPAMD_KV_SPLIT=$([ $path = f32 ] && echo 0 || echo 1) timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${path}_t${t}_$name -o pmc -- python $R/scripts/kv_only.py 500000 $t 2 > $R/$OUT/pmc_${path}_t${t}_$name.log 2>&1); echo "pmc $path t=$t $name rc=$?"; }
for cfg in f32:65 split:65; do
  path=${cfg%%:*}; t=${cfg##*:}
  pmc $path $t mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
  pmc $path $t insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  pmc $path $t fetch FETCH_SIZE
  pmc $path $t write WRITE_SIZE
done
find $OUT -name "*kernel_trace*" -size +5M -delete
python scripts/collect_profiles.py $TAG $ROUND
cp profiles/kv_pmc_current.json profiles/kv_pmc_split_current.json profiles/r${ROUND}_${TAG}_kv_pmc_*.json $OUT/ 2>/dev/null
