#!/bin/bash
# round 5, GPU session 5: 128-row (medium) blocks + saturated norms, float64 lean generation + VALU contraction, f64 MFMA / VALU overlap, workloads again
set +e
OUT=gpurun_out/r5s5; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl mfma_f64_valu_overlap.hip > /dev/null 2>&1 && timeout 120 /tmp/ovl > $R/$OUT/0_f64_overlap.txt 2>&1); echo "[f64 overlap] rc=$?"; cat $OUT/0_f64_overlap.txt | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_recenter.py -m gpu -q -x > $OUT/1_recenter.log 2>&1; echo "[recenter] rc=$?"; tail -5 $OUT/1_recenter.log
timeout 400 python -m pytest tests/test_gpu_generic.py -m gpu -q > $OUT/2_generic.log 2>&1; echo "[generic f64] rc=$?"; tail -5 $OUT/2_generic.log
timeout 120 python scripts/f64_gen_timing.py $OUT/3_f64_gen_timing.json > $OUT/3_f64_gen_timing.log 2>&1; echo "[f64 timing] rc=$?"; tail -10 $OUT/3_f64_gen_timing.log | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_road -o road -- python $R/scripts/workload_breakdown.py road3d plain > $R/$OUT/5_road_plain.log 2>&1); echo "[road plain rocprof] rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_protein -o protein -- python $R/scripts/workload_breakdown.py protein plain > $R/$OUT/6_protein_plain.log 2>&1); echo "[protein plain rocprof] rc=$?"
python - <<'PY'
import json, re
for f in ("gpurun_out/r5s5/5_road_plain.log", "gpurun_out/r5s5/6_protein_plain.log"):
    try:
        s = open(f).read(); j = json.loads(s[s.index("{"):s.rindex("}") + 1])
        for st in j["stages"]: print(st["stage"][:40], [round(x, 4) for x in st["seconds_per_iteration"]], st["cg_iterations"], st.get("kernel_path"), st.get("rows_by_region"))
    except Exception as e: print(f, "unparsed", e)
PY
for f in $(find $OUT/prof_road $OUT/prof_protein -name "*kernel_stats*.csv"); do echo $f; head -8 $f | cut -c1-160; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 500 python -m pytest tests/test_gpu_c5_at_size.py tests/test_gpu_kv.py tests/test_gpu_kv_split.py tests/test_gpu_grad2.py -m gpu -q -x > $OUT/7_tests.log 2>&1; echo "[c5 + kv + kv_split + grad2] rc=$?"; tail -4 $OUT/7_tests.log
