#!/bin/bash
# Dump the ISA of one kernel of a translation unit: scripts/isa_loop.sh <file.hip> <mangled-name-substring> [out.s]
cd "$(dirname "$0")/../gpytorch_amd/csrc" || exit 1
out=${3:-/tmp/isa_kernel.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I. -I../../include -S --cuda-device-only "$1" -o /tmp/isa_all.s 2>/dev/null
awk -v pat="$2" 'index($0, "_Z") == 1 && index($0, pat) && /:/ {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/isa_all.s > "$out"
wc -l "$out"
