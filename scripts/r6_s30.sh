#!/bin/bash
# round 6, GPU session 30: the tune library rebuilt on the ONE loop body (kv_gramh_body.inc): every ablation / geometry case of scripts/kgh_ablate.py
# runs; the product kernel timed beside ABL = 0 of the tune build; hazard stress test (tune SAFE builds); split kernel sweep
set +e
OUT=gpurun_out/r6s30; mkdir -p $OUT
timeout 300 python scripts/kgh_ablate.py r6s30 500000 "0:2,1:2,2:2,3:2,4:2,5:2,6:2,7:2,8:2,10:2,0:4,0:2:1,0:4:1,110:2,120:1,130:4" > $OUT/1_ablate.log 2>&1; echo "[kgh_ablate] rc=$?"; tail -20 $OUT/1_ablate.log | cut -c1-220
cp gpurun_out/kgh_ablate_r6s30.json $OUT/ 2>/dev/null
timeout 300 python scripts/kv_split_time.py > $OUT/2_kv_split_time.log 2>&1; echo "[kv_split_time] rc=$?"; tail -6 $OUT/2_kv_split_time.log | cut -c1-220
timeout 400 python -m pytest tests/test_gpu_hazard_stress.py tests/test_gpu_kv_split.py -m gpu -q -x > $OUT/3_tests.log 2>&1; echo "[hazard stress + kv_split tests] rc=$?"; tail -3 $OUT/3_tests.log
