#!/bin/bash
set +e
OUT=gpurun_out/r2s23; mkdir -p $OUT
for rep in 1 2; do
for lib in libgpamd.so libgpamd_noprio.so; do
  for t in 65 33; do
    GPAMD_KV_SPLIT=0 GPAMD_LIBRARY=$PWD/gpytorch_amd/csrc/$lib timeout 200 python scripts/kv_time.py 500000 $t 4 2>&1 | grep -E "^\{" | sed "s/^/$lib /"
  done
done
done | tee $OUT/setprio_ab.txt
