#!/bin/bash
# Session 3: parity tests + kernel-variant tuning + PMC counters for the dominant kernel.
set +e
TAG=${1:-s3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | tail -15
echo "== tune"; timeout 900 python scripts/kv_tune.py 100000 5 > $OUT/tune.log 2>&1; echo "tune rc=$?"; tail -25 $OUT/tune.log
echo "== counters list"; (cd /tmp && timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters.txt 2>&1); grep -ciE "mfma" $OUT/counters.txt
grep -oE "(SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|FETCH_SIZE|WRITE_SIZE|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|SQ_LDS_BANK_CONFLICT|SQ_INST_CYCLES_VMEM|MfmaUtil|VALUBusy)" $OUT/counters.txt | sort | uniq | tr '\n' ' '
pmc() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o pmc -- python $GRAFT_REPO_ROOT/scripts/kv_only.py 100000 65 3 > $GRAFT_REPO_ROOT/$OUT/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
find $OUT -name "*counter_collection*.csv" | head
find $OUT -name "*kernel_trace*" -size +5M -delete
