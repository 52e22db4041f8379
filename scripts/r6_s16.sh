#!/bin/bash
# round 6, GPU session 16: KV_SPLIT_FEW (culled products of fewer than five columns on the split kernels): tests, road3d with the cutoff (prediction time)
set +e
OUT=gpurun_out/r6s16; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_far_cull.py tests/test_gpu_kv_split.py tests/test_gpu_recenter.py tests/test_gpu_model.py -q -m gpu -x > $OUT/0_tests.log 2>&1; echo "[tests] rc=$?"; tail -4 $OUT/0_tests.log
timeout 300 python bench.py --config road3d --far-cutoff 1e-7 > $OUT/1_bench_road3d_far.json 2> $OUT/1_bench_road3d_far.err; echo "[bench road3d far 1e-7] rc=$?"; python - <<'PY'
import json
c=json.loads(open('gpurun_out/r6s16/1_bench_road3d_far.json').read().strip().splitlines()[-1])['config']
print({k:c[k] for k in ('seconds_per_iteration_median','training_seconds','prediction_seconds_cold_caches','test_rmse','variance_min')}, [round(v,3) for v in c['seconds_per_iteration']])
PY
