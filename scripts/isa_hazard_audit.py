"""Static audit of the SHIPPED gfx950 code objects: how many wait states lie between every matrix-pipe instruction that writes VGPRs and the
first non-matrix instruction that reads one of those registers.

Why: on gfx950 a VALU read of a VGPR that a preceding v_mfma is still writing is NOT interlocked, and the compiler's hazard table
(s_nop 11 after v_mfma_f32_32x32x16_f16) was measured insufficient on a full chip.  The kernels that consume MFMA results
on the VALU (kvs_*, kvm_* translation units, -amdgpu-mfma-vgpr-form=1) therefore put `mfma_result_fence(regs...)` (32 wait states, tied to
the result registers by data dependencies -- common.hpp) between producer and reader.  Whether the fence is still where it must be after
the optimiser has had its way (round 3 lost two MFMAs BELOW a fence that carried no data dependency) can only be seen in the ISA, so this
script disassembles libgpamd.so itself and measures it:

  for every v_mfma* with a VGPR destination: wait states (instructions issued; s_nop N counts N + 1) until the first non-MFMA instruction
  that reads any destination register, following fall-through order and -- for loop back-edges -- the loop head.

Output: one line per (kernel, MFMA opcode) with the minimum distance; exit status 1 if any VGPR-form kernel (kv_gramv / kv_gram4 / kv_gram16 /
kv_grad2) has a distance below --min (default 32).  tests/test_isa_hazard_cpu.py runs it on the built library.
Usage: python scripts/isa_hazard_audit.py [--lib path] [--min 32] [--json out.json] [--all]"""
from __future__ import annotations

import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FENCED = ("kv_gramv_kernel", "kv_gram4_kernel", "kv_gram16_kernel", "kv_grad2_kernel")
COMPILER_TABLE = 12   # wait states the toolchain itself guarantees after an 8-pass XDL write (s_nop 11)
# per-family bars above the toolchain's table (round 5): the headline fp32-contraction kernel carries 8 explicit wait states behind its Gram MFMAs
# (kv_gram.hpp) -- 20 in all; round 6: kv_gramh_kernel (the library-default split kernel) carries the same 8 tied wait states behind its Gram MFMAs
# (the full fence was measured at 1.0 - 1.4 %: profiles/r06_s4_kv_gramh_fence_ab.json) and stays covered by the on-device stress test against its
# fully fenced build (tests/test_gpu_hazard_stress.py)
MIN_BY_FAMILY = {"kv_gram_kernel": 20, "kv_gramh_kernel": 20}


def required(fam: str, fenced_min: int = 32) -> int:
    return fenced_min if fam in FENCED else MIN_BY_FAMILY.get(fam, COMPILER_TABLE)


def extract_code_objects(lib_path: str, arch: str = "gfx950"):
    """Yield the device ELF images of every offload bundle embedded in the shared library."""
    blob = open(lib_path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (num,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(num):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24 : off + 24 + tl].decode()
            off += 24 + tl
            if arch in triple and size:
                yield blob[pos + o : pos + o + size]
        pos += len(MAGIC)


REG_RE = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(operand: str):
    """VGPR numbers named by one operand string (v3, v[2:17]); AGPRs are returned offset by 1000."""
    out = []
    for m in REG_RE.finditer(operand):
        if m.group(1):
            base, a, b = m.group(1), int(m.group(2)), int(m.group(2))
        else:
            base, a, b = m.group(3), int(m.group(4)), int(m.group(5))
        off = 1000 if base == "a" else 0
        out.extend(range(a + off, b + 1 + off))
    return out


def split_operands(text: str):
    depth, cur, out = 0, "", []
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


# instructions whose FIRST operand is a source, not a destination
STORE_LIKE = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_add", "ds_max", "ds_min",
              "s_", "v_cmp", "v_cmpx", "buffer_atomic", "flat_atomic", "v_readfirstlane", "v_readlane")


def parse_function(lines):
    """[(addr, mnemonic, dst regs, src regs, wait states, branch target or None)]"""
    ins = []
    for ln in lines:
        m = re.match(r"\s*([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if not m:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        ops = re.sub(r"\b(offset|offen|idxen|glc|slc|nt|sc0|sc1|lds|off|cbsz|abid|blgp|op_sel|op_sel_hi|neg_lo|neg_hi|clamp|mul|div|row_\w+|quad_perm|bank_mask|row_mask|bound_ctrl|dst_sel|src0_sel|src1_sel|dst_unused)\b[:\[\]0-9,x]*", "", ops)
        opl = split_operands(ops)
        target = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = re.search(r"<[^>]*\+0x([0-9A-Fa-f]+)>", ln)
            target = int(t.group(1), 16) if t else -1
        ws = 1
        if mn == "s_nop":
            ws = int(opl[0], 0) + 1 if opl else 1
        if mn.startswith(STORE_LIKE) or not opl:
            dst, src = [], [r for o in opl for r in regs_of(o)]
        else:
            dst, src = regs_of(opl[0]), [r for o in opl[1:] for r in regs_of(o)]
            if mn.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot")) or mn.endswith(("_fmac_f32",)):
                src += dst   # accumulating forms read their destination
        ins.append((addr, mn, dst, src, ws, target))
    return ins


def audit_function(name, ins, horizon=96):
    """{mfma opcode: (min distance, address of the closest reader)} over VGPR-destination MFMAs; distance capped at `horizon` (= far enough)."""
    base = ins[0][0] if ins else 0
    index_of = {a: k for k, (a, *_rest) in enumerate(ins)}
    res = {}

    def scan(start_k, regs, dist0, seen_back):
        """Walk forward from instruction index start_k; returns (distance to first non-MFMA reader, reader address) or (horizon, None)."""
        d = dist0
        live = set(regs)
        k = start_k
        while k < len(ins) and d < horizon and live:
            addr, mn, dst, src, ws, target = ins[k]
            is_mfma = mn.startswith(("v_mfma", "v_smfmac"))
            if not is_mfma and live.intersection(src):
                return d, addr
            if not is_mfma:
                live.difference_update(dst)      # overwritten by something else: no longer the MFMA's value
            elif live.intersection(dst):
                return horizon, None             # a later MFMA rewrites the registers: that one is audited on its own
            if target is not None and target >= 0:
                tk = index_of.get(base + target)
                if tk is not None and tk <= k and (k, tk) not in seen_back:   # back-edge: also follow the loop head
                    bd, ba = scan(tk, live, d + ws, seen_back | {(k, tk)})
                    if bd < horizon:
                        fd, fa = scan(k + 1, live, d + ws, seen_back)
                        return (bd, ba) if bd <= fd else (fd, fa)
                if mn == "s_branch":
                    if tk is not None and tk > k:
                        k = tk
                        d += ws
                        continue
                    return horizon, None
            if mn == "s_endpgm":
                break
            d += ws
            k += 1
        return horizon, None

    for k, (addr, mn, dst, src, ws, target) in enumerate(ins):
        if not mn.startswith(("v_mfma", "v_smfmac")) or not dst or dst[0] >= 1000:
            continue
        d, ra = scan(k + 1, dst, 0, frozenset())
        cur = res.get(mn)
        if cur is None or d < cur[0]:
            res[mn] = (d, ra, addr)
    return res


def demangle_all(syms):
    """{mangled: 'kernel<args>'} through one c++filt process (binutils); falls back to the mangled names."""
    try:
        out = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        out = list(syms)
    res = {}
    for sym, d in zip(syms, out):
        d = re.sub(r"^void\s+", "", d)
        res[sym] = re.sub(r"\(.*$", "", d).replace("gpamd::", "")
    return res


def _audit_one(path):
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    cur, buf = None, []
    funcs = {}
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-fA-F]+ <([^>]+)>:", ln)
        if m:
            if cur:
                funcs[cur] = buf
            cur, buf = m.group(1), []
        elif cur:
            buf.append(ln)
    if cur:
        funcs[cur] = buf
    rows = []
    for sym, lines in funcs.items():
        if not any("v_mfma" in ln for ln in lines):
            continue
        ins = parse_function(lines)
        for op, (d, ra, ma) in audit_function(sym, ins).items():
            rows.append({"kernel": sym, "mfma": op, "min_wait_states": d, "mfma_addr": hex(ma), "reader_addr": hex(ra) if ra else None})
    return rows


def audit_library(lib_path: str):
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for n, img in enumerate(extract_code_objects(lib_path)):
            p = os.path.join(tmp, f"co{n}.elf")
            open(p, "wb").write(img)
            paths.append(p)
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            rows = [r for part in ex.map(_audit_one, paths) for r in part]
    names = demangle_all(sorted({r["kernel"] for r in rows}))
    for r in rows:
        r["kernel"] = names[r["kernel"]]
    return rows


def family(kernel: str) -> str:
    return kernel.split("<")[0]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(here, "..", "gpytorch_amd", "csrc", "libgpamd.so"))
    ap.add_argument("--min", type=int, default=32)
    ap.add_argument("--json", default=None)
    ap.add_argument("--all", action="store_true", help="print every kernel, not only the offenders and the per-family minima")
    a = ap.parse_args()
    rows = audit_library(a.lib)
    # fenced families (MFMA results consumed by the VALU behind mfma_result_fence): >= --min; every other family: at least the compiler's own
    # hazard-table distance for an 8-pass MFMA (s_nop 11 -> 12 wait states), i.e. no reader may ever sit closer than the toolchain promises
    bad = [r for r in rows if r["min_wait_states"] < (a.min if family(r["kernel"]) in FENCED else COMPILER_TABLE)]
    fam = {}
    for r in rows:
        key = (family(r["kernel"]), r["mfma"])
        if key not in fam or r["min_wait_states"] < fam[key]["min_wait_states"]:
            fam[key] = r
    print(f"{len(rows)} (kernel, VGPR-destination MFMA opcode) pairs in {a.lib}")
    for (k, op), r in sorted(fam.items()):
        print(f"  {k:20s} {op:30s} min wait states {r['min_wait_states']:3d}  {'fenced' if k in FENCED else '      '}  ({r['kernel'][:80]})")
    if a.all:
        for r in sorted(rows, key=lambda r: r["min_wait_states"]):
            print(r)
    for r in bad:
        print("BELOW THE REQUIRED DISTANCE:", r)
    if a.json:
        json.dump({"min_required_fenced": a.min, "min_required_other": COMPILER_TABLE, "pairs": len(rows),
                   "families": [dict(family=k, mfma=op, fenced=k in FENCED, min_wait_states=r["min_wait_states"], closest=r["kernel"]) for (k, op), r in sorted(fam.items())],
                   "offenders": bad}, open(a.json, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
