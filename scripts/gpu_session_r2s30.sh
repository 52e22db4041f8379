#!/bin/bash
set +e
OUT=gpurun_out/r2s30; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python scripts/cg_graph_timing.py r2s30 > $OUT/timing.log 2>&1; echo "rc=$?"; grep -E "^\{|Error|error|Warn" $OUT/timing.log | cut -c1-230
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o cg -- python $R/scripts/cg_graph_timing.py r2s30p > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -24 $f | cut -c1-150; done
find $OUT -name "*kernel_trace*" -size +5M -delete
