#!/bin/bash
# round 5, GPU session 9: PMC passes of the round's new kernels (kv_directh at the road3d shape; the float64 kernels after the lean generation)
set +e
OUT=gpurun_out/r5s9; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pmc() { case=$1; name=$2; shift; shift; cmd=$1; shift; (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${case}_$name -o pmc -- $cmd > $R/$OUT/pmc_${case}_$name.log 2>&1); echo "pmc $case $name rc=$?"; }
pmc directh_t11 mfma "python $R/scripts/kv_direct_only.py 217437 11 3" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc directh_t11 insts "python $R/scripts/kv_direct_only.py 217437 11 3" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc f64_t65 mfma "python $R/scripts/kv_f64_only.py 100000 3 65 3" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc f64_t65 insts "python $R/scripts/kv_f64_only.py 100000 3 65 3" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc f64_t1 mfma "python $R/scripts/kv_f64_only.py 100000 3 1 3" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc f64_t1 insts "python $R/scripts/kv_f64_only.py 100000 3 1 3" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
timeout 120 python scripts/kv_gram_fence_ab.py $OUT/kv_gram_fence_ab.json > $OUT/kv_gram_fence_ab.log 2>&1; echo "[fence A/B] rc=$?"; cat $OUT/kv_gram_fence_ab.log | cut -c1-600
find $OUT -name "*kernel_trace*" -size +5M -delete
python scripts/pmc_collect_r5.py $OUT $OUT/kv_pmc
