#!/bin/bash
# round 5, GPU session 6: direct differences + split contraction (kv_directh): parity, timing, the workloads again; hazard stress (tune library now built by build())
set +e
OUT=gpurun_out/r5s6; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_kv_split.py tests/test_gpu_kv.py tests/test_gpu_recenter.py tests/test_gpu_hazard_stress.py -m gpu -q -x > $OUT/1_tests.log 2>&1; echo "[kv_split + kv + recenter + hazard] rc=$?"; tail -5 $OUT/1_tests.log
timeout 200 python scripts/kv_direct_split_timing.py $OUT/2_direct_split_timing.json > $OUT/2_direct_split_timing.log 2>&1; echo "[direct split timing] rc=$?"; tail -9 $OUT/2_direct_split_timing.log | cut -c1-330
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_road -o road -- python $R/scripts/workload_breakdown.py road3d plain > $R/$OUT/5_road_plain.log 2>&1); echo "[road plain rocprof] rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_protein -o protein -- python $R/scripts/workload_breakdown.py protein plain > $R/$OUT/6_protein_plain.log 2>&1); echo "[protein plain rocprof] rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5s6/5_road_plain.log", "gpurun_out/r5s6/6_protein_plain.log"):
    try:
        s = open(f).read(); j = json.loads(s[s.index("{"):s.rindex("}") + 1])
        for st in j["stages"]: print(st["stage"][:40], [round(x, 4) for x in st["seconds_per_iteration"]], st["cg_iterations"], st.get("kernel_path"), st.get("rows_by_region"))
    except Exception as e: print(f, "unparsed", e)
PY
for f in $(find $OUT/prof_road $OUT/prof_protein -name "*kernel_stats*.csv"); do echo $f; head -8 $f | cut -c1-160; done
find $OUT -name "*kernel_trace*" -size +5M -delete
timeout 400 python -m pytest tests/test_gpu_bbmm.py tests/test_gpu_model.py tests/test_gpu_extra.py -m gpu -q -x > $OUT/7_tests.log 2>&1; echo "[bbmm model extra] rc=$?"; tail -4 $OUT/7_tests.log
