#!/bin/bash
# Session 4: parity tests + tuning sweep incl. ablations.
set +e
TAG=${1:-s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | tail -15
echo "== tune"; timeout 900 python scripts/kv_tune.py 100000 5 > $OUT/tune.log 2>&1; echo "tune rc=$?"; tail -30 $OUT/tune.log
