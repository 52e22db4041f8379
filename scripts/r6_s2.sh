#!/bin/bash
# round 6, GPU session 2: bench extras (MLL per preconditioner rank, posterior per setting) at the metric shape on the library defaults
set +e
OUT=gpurun_out/r6s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 1 --warmup 0 --other-steps 0 --skip-cpu-baseline --skip-parity > $OUT/1_bench_metric_extras.json 2> $OUT/1_bench.err; echo "[bench extras] rc=$?"; tail -5 $OUT/1_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6s2/1_bench_metric_extras.json").read().strip().splitlines()[-1])
e = d["extras"]
for r in e["mll_by_preconditioner_rank"]: print(r)
for r in e["posterior"]: print(r)
PY
