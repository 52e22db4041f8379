#!/bin/bash
# One parameterised GPU session (replaces the per-session scripts of rounds 1-2): `scripts/gpu_run.sh <tag> <step> [<step> ...]`.
# Every step runs under its own timeout, writes to gpurun_out/<tag>/ and never aborts the session.
#   tests[:expr]      pytest -m gpu (optionally -k expr)         file:<path>     pytest on one file
#   bench:<args>      python bench.py <args> (commas -> spaces)  prof:<args>     rocprofv3 --kernel-trace --stats of bench.py <args>
#   py:<script+args>  python <script> <args> (commas -> spaces)
set +e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for step in "$@"; do
  i=$((i+1))
  kind=${step%%:*}; arg=${step#*:}; [ "$kind" = "$step" ] && arg=""
  arg=${arg//,/ }
  case $kind in
    tests) timeout 900 python -m pytest tests -m gpu -q -x ${arg:+-k "$arg"} > $OUT/${i}_tests.log 2>&1; echo "[$step] rc=$?"; tail -5 $OUT/${i}_tests.log ;;
    file)  timeout 900 python -m pytest $arg -m gpu -q > $OUT/${i}_file.log 2>&1; echo "[$step] rc=$?"; tail -8 $OUT/${i}_file.log ;;
    bench) timeout 900 python bench.py $arg > $OUT/${i}_bench.json 2> $OUT/${i}_bench.err; echo "[$step] rc=$?"; cut -c1-1500 $OUT/${i}_bench.json; tail -3 $OUT/${i}_bench.err ;;
    prof)  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$i -o prof -- python $GRAFT_REPO_ROOT/bench.py $arg > $GRAFT_REPO_ROOT/$OUT/${i}_prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/${i}_prof.err); echo "[$step] rc=$?"; find $OUT/prof_$i -name "*kernel_stats.csv" | head -1 | xargs -r head -8 ;;
    py)    timeout 900 python $arg > $OUT/${i}_py.log 2>&1; echo "[$step] rc=$?"; tail -25 $OUT/${i}_py.log ;;
    *) echo "unknown step $step" ;;
  esac
done
