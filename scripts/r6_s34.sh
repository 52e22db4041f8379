#!/bin/bash
# round 6, GPU session 34: does the 4.35 s accurate cold posterior of session 32's bench extras reproduce?  (standalone: 0.87 s)
set +e
OUT=gpurun_out/r6s34; mkdir -p $OUT
for rep in 1; do
timeout 600 python bench.py --steps 1 --warmup 0 --skip-split --skip-parity --skip-cpu-baseline > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err; echo "[bench extras $rep] rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_$rep.json"))["extras"]
print([(p["settings"], round(p["cold_ms"]), round(p["warm_ms"], 1), [round(x) for x in p["cold_ms_all"]]) for p in d["posterior"]])
PY
done
