#!/bin/bash
set +e
OUT=gpurun_out/r2s4; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_grad2.py tests/test_gpu_extra.py tests/test_gpu_bbmm.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | head -40
timeout 900 python scripts/grad_timing.py r2s4 > $OUT/grad_timing.log 2>&1; echo "grad rc=$?"; python - <<'PY'
import json
for r in json.load(open("gpurun_out/grad_timing_r2s4.json")):
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
PY
timeout 900 python bench.py --steps 1 --warmup 1 --skip-extras > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?"; cat $OUT/bench_quick.json; tail -3 $OUT/bench_quick.err
GPAMD_BENCH_BACKEND=gloo GPAMD_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --config c4 --size 100000 --probes 64 > $OUT/bench_2rank_c4small.json 2> $OUT/bench_2rank.err; echo "bench2 rc=$?"; cat $OUT/bench_2rank_c4small.json; tail -3 $OUT/bench_2rank.err
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from gpytorch_amd import backend as B
dev = torch.device("cuda:0"); n = 500_000
X = torch.rand(n, 3, device=dev); xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
for t in (11, 16, 17):
    vt = torch.randn(t, B.round_up(n, 4), device=dev); B.kv(xp, xp, vt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(3): B.kv(xp, xp, vt)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 3
    print(f"t={t}: {ms:.1f} ms  {2.0*n*n*t/ms/1e9:.1f} TF  frac {2.0*n*n*t/ms/1e9/157.3:.3f}")
PY
