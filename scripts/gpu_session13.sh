#!/bin/bash
set +e
OUT=gpurun_out/s13; mkdir -p $OUT
timeout 600 python scripts/kv_tune.py 100000 5 26,27,28,29 > $OUT/tune.log 2>&1; tail -5 $OUT/tune.log
timeout 600 python scripts/kv_tune.py 500000 2 26,28 > $OUT/tune500k.log 2>&1; tail -3 $OUT/tune500k.log
