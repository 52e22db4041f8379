#!/bin/bash
set +e
OUT=gpurun_out/r2s8; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_extra.py tests/test_gpu_compose.py tests/test_gpu_bbmm.py tests/test_gpu_parity_at_size.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -40
timeout 1500 python bench.py --steps 1 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2s8/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("parity"), d.get("extras"), d.get("cpu_baseline"))
PY
tail -3 $OUT/bench.err
