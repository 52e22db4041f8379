#!/bin/bash
# GPU-box session: the -m gpu suite (with durations) + smoke.  Usage: gpurun -- 'bash scripts/gpu_tests.sh <tag> [pytest args]'
set +e
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
( nproc; free -g | head -2; rocm-smi --showmeminfo vram | head -8 ) > $OUT/env.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^[0-9.]+s (call|setup)" $OUT/pytest.log | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
