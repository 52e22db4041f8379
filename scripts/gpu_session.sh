#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything is logged
# under gpurun_out/ (merged back by gpurun).  Usage: gpurun -- 'bash scripts/gpu_session.sh [tag]'
set +e
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( rocm-smi --showproductname 2>/dev/null | head -20; nproc; free -g | head -2 ) > $OUT/env.log 2>&1
echo "== pytest" ; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --skip-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
# keep the merged-back payload small
find $OUT/prof -name "*kernel_trace*" -size +20M -delete
