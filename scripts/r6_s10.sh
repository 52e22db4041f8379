#!/bin/bash
# round 6, GPU session 10: the auto rules (preconditioner 256 at n >= 200 000, Lanczos block 32 at n >= 262 144) -- block-Lanczos tests on the native b = 32
# kernels, bench extras with them, PMC passes of the two 65-column kernels on the final binary
set +e
TAG=r6s10; ROUND=06
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_love_vs_oracle.py tests/test_gpu_model.py tests/test_gpu_extra.py -m gpu -q -k "love or lanczos or block or fantasy or posterior" > $OUT/1_tests.log 2>&1; echo "[love / lanczos tests] rc=$?"; tail -6 $OUT/1_tests.log
timeout 300 python scripts/love_block_timing.py c2 > $OUT/2_love_block_c2.log 2>&1; echo "[love block timing c2] rc=$?"; tail -12 $OUT/2_love_block_c2.log | cut -c1-200
timeout 600 python bench.py --steps 1 --warmup 0 --other-steps 0 --skip-cpu-baseline --skip-parity > $OUT/3_bench_metric_extras.json 2> $OUT/3_bench.err; echo "[bench extras] rc=$?"; tail -3 $OUT/3_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6s10/3_bench_metric_extras.json").read().strip().splitlines()[-1])
e = d["extras"]
for r in e["mll_by_preconditioner_rank"]: print(r)
for r in e["posterior"]: print(r)
PY
pmc() { path=$1; t=$2; name=$3; shift; shift; shift; (cd /tmp && GPAMD_KV_SPLIT=$([ $path = f32 ] && echo 0 || echo 1) timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${path}_t${t}_$name -o pmc -- python $R/scripts/kv_only.py 500000 $t 2 > $R/$OUT/pmc_${path}_t${t}_$name.log 2>&1); echo "pmc $path t=$t $name rc=$?"; }
for cfg in f32:65 split:65; do
  path=${cfg%%:*}; t=${cfg##*:}
  pmc $path $t mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
  pmc $path $t insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  pmc $path $t fetch FETCH_SIZE
  pmc $path $t write WRITE_SIZE
done
find $OUT -name "*kernel_trace*" -size +5M -delete
python scripts/collect_profiles.py $TAG $ROUND
cp profiles/kv_pmc_current.json profiles/kv_pmc_split_current.json profiles/r${ROUND}_${TAG}_kv_pmc_*.json $OUT/ 2>/dev/null
