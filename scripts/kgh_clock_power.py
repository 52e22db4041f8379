"""Is the split-operand K*V kernel (kv_gramh) power-limited or issue-limited?  Each ablation build of libgpamd_tune.so (full loop / no
generation VALU / no contraction MFMAs / one wave per SIMD) and the fp32-MFMA product kernel run back to back for a few seconds while the
shader clock and the package power are sampled (rocm-smi --showclocks --showpower, 4 Hz): if the clock rises when either pipe idles, the
full kernel is power-limited; if it does not, it is issue-limited.
    python scripts/kgh_clock_power.py [tag] [n] [seconds per case] -> gpurun_out/kgh_clock_power_<tag>.json"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpytorch_amd import backend as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
dev = torch.device("cuda:0")


class Sampler:
    def __init__(self):
        self.rows, self._stop, self._t = [], threading.Event(), None

    def _loop(self):
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r"GPU\[0\]\s*:\s*sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
                p = re.search(r"GPU\[0\]\s*:\s*[^\n]*[Pp]ower[^\n]*?:\s*([0-9.]+)", out)
                if c:
                    self.rows.append((int(c.group(1)), float(p.group(1)) if p else None))
            except Exception:
                pass
            self._stop.wait(0.25)

    def __enter__(self):
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=6)

    def summary(self):
        rows = self.rows[2:] if len(self.rows) > 4 else self.rows          # the first samples still see the previous case
        if not rows:
            return {"samples": 0}
        ck = sorted(r[0] for r in rows)
        pw = sorted(r[1] for r in rows if r[1] is not None)
        return {"samples": len(rows), "sclk_mhz_median": ck[len(ck) // 2], "sclk_mhz_min": ck[0], "sclk_mhz_max": ck[-1],
                "power_w_median": pw[len(pw) // 2] if pw else None, "power_w_max": pw[-1] if pw else None}


h = C.CDLL(os.path.join(ROOT, "gpytorch_amd", "csrc", "libgpamd_tune.so"))
f = h.gpamd_tune_kv_gramh_rbf3
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
              C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(0)
X = torch.rand(n, 3, device=dev)
xp = B.prep_points("rbf", X, torch.tensor(0.25), X.mean(0))
ld = B.round_up(n, 4)
ldh = (n + 127) // 128 * 128
V = torch.randn(65, ld, device=dev)
Vh = torch.randn(64, ldh, device=dev).half()
Vl = (1e-3 * torch.randn(64, ldh, device=dev)).half()
colmul = torch.ones(80, device=dev)
S, jc, _ = B.kv_plan("rbf", n, n, 3, 64, B.KV_GRAM | B.KV_SPLIT, ld)
P = torch.empty(S * 65 * ld, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tune(abl, ni):
    def run():
        rc = f(abl, ni, 0, xp.xp.data_ptr(), n, xp.xp.data_ptr(), n, V.data_ptr(), ld, Vh.data_ptr(), Vl.data_ptr(), ldh, colmul.data_ptr(),
               P.data_ptr(), ld, S, jc, st)
        assert rc == 0, rc
    return run


def product(split):
    V64 = V[:64].contiguous()

    def run():
        B.SPLIT_CONTRACTION = split
        B.kv(xp, xp, V64)
    return run


cases = [("idle", None), ("kv_gramh full loop (tune build, 2 row tiles)", tune(0, 2)), ("kv_gramh without the generation VALU", tune(1, 2)),
         ("kv_gramh without the contraction MFMAs", tune(2, 2)), ("kv_gramh, one wave per SIMD", tune(7, 2)),
         ("kv_gramh, V planes staged once (no LDS re-staging)", tune(3, 2)), ("kv_gramh, staged once and no barriers", tune(4, 2)),
         ("kv_gramh, A operands from one block (no LDS operand traffic growth)", tune(5, 2)),
         ("product kernel, split contraction (4 row tiles)", product(True)), ("product kernel, fp32 MFMA contraction", product(False))]
out = {"n": n, "columns": 64, "seconds_per_case": secs, "cases": []}
for name, run in cases:
    if run is None:
        with Sampler() as s:
            time.sleep(secs)
        out["cases"].append({"case": name, **s.summary()})
        print(json.dumps(out["cases"][-1]), flush=True)
        continue
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k = 0
    with Sampler() as s:
        t0 = time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(4):
                run()
            k += 4
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
    rec = {"case": name, "launches": k, "ms_per_launch": e0.elapsed_time(e1) / k, **s.summary()}
    out["cases"].append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/kgh_clock_power_{tag}.json", "w"), indent=1)
