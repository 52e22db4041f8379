#!/bin/bash
# Instruction-order picture of the hottest loop of one kernel: M = MFMA (new line), v = VALU, D = LDS, w = s_waitcnt, n = s_nop
# usage: scripts/isa_view.sh <file.hip> <mangled-substring> [extra hipcc flags...]
cd "$(dirname "$0")/../gpytorch_amd/csrc" || exit 1
f=$1; pat=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I. -I../../include -S --cuda-device-only "$@" "$f" -o /tmp/isa_all.s 2>&1 | grep -E "error" 
awk -v pat="$pat" 'index($0, "_Z") == 1 && index($0, pat) && /:/ {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/isa_all.s > /tmp/isa_kernel.s
python3 - <<'PY'
import re
lines = open("/tmp/isa_kernel.s").read().split("\n")
# innermost loop = the backward branch whose body holds the most MFMAs
labels = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = lines[labels[m.group(1)]:i + 1]
        n = sum("v_mfma" in b for b in body)
        if best is None or (n > 0 and len(body) < len(best[1]) and n >= best[0]) or n > best[0]:
            if best is None or n >= best[0]: best = (n, body)
n, body = best
out = []; cnt = {}
for b in body:
    t = b.split()
    if not t or t[0].startswith(";") or t[0].endswith(":"): continue
    op = t[0]; cnt[op] = cnt.get(op, 0) + 1
    if "mfma" in op: out.append("\nM ")
    elif op.startswith("v_"): out.append("v")
    elif op.startswith("ds_"): out.append("D")
    elif op == "s_waitcnt": out.append("w")
    elif op == "s_nop": out.append("n")
    elif op.startswith("scratch_"): out.append("S")
    elif op.startswith("global_") or op.startswith("buffer_"): out.append("G")
    else: out.append(".")
print("".join(out))
print(sorted(cnt.items(), key=lambda kv: -kv[1])[:12])
PY
