#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats + PMC passes for the
# dominant kernel.  Everything is logged under gpurun_out/<tag>/ (merged back by gpurun).
# Usage: gpurun -- 'bash scripts/gpu_session.sh <tag> [skip-tests]'
set +e
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( nproc; free -g | head -2 ) > $OUT/env.log 2>&1
if [ "$2" != "skip-tests" ]; then
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | tail -15
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
fi
echo "== bench"; timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprof stats"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --skip-cpu-baseline --skip-extras > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-220; done
pmc() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_$name -o pmc -- python $R/scripts/kv_only.py 500000 65 2 > $R/$OUT/pmc_$name.log 2>&1); echo "pmc $name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
find $OUT -name "*kernel_trace*" -size +5M -delete
