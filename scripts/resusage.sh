#!/bin/bash
# Per-kernel register / spill / LDS summary of one translation unit: scripts/resusage.sh <file.hip> [extra hipcc flags]
cd "$(dirname "$0")/../gpytorch_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I. -I../../include -c "$f" -o /tmp/resusage.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
name = None; rec = {}
for line in sys.stdin:
    if "error" in line or "warning" in line: print(line.rstrip())
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        name = v; rec = {}
    else:
        rec[k] = v
        if k.startswith("LDS"):
            print(name[:70].ljust(72), "vgpr", rec.get("VGPRs"), "agpr", rec.get("AGPRs"), "spill", rec.get("VGPRs Spill"), "occ", rec.get("Occupancy [waves/SIMD]"), "lds", v)
'
