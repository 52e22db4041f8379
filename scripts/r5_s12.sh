#!/bin/bash
# round 5, GPU session 12: the final auto rule of settings.lanczos_block_size (block Lanczos from rank 200 on): at-size posterior tests, bench extras
set +e
OUT=gpurun_out/r5s12; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_dense_at_size.py -m gpu -q -k posterior > $OUT/1_posterior.log 2>&1; echo "[posterior at size] rc=$?"; tail -3 $OUT/1_posterior.log
timeout 200 python bench.py --steps 1 --warmup 0 --other-steps 0 --skip-cpu-baseline --skip-parity > $OUT/2_bench_metric_extras.json 2> $OUT/2_bench.err; echo "[bench extras] rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5s12/2_bench_metric_extras.json").read().strip().splitlines()[-1])
print({k: d["extras"][k] for k in d["extras"] if k.startswith("posterior") or k.startswith("mll")})
PY
