"""Static guard for the MFMA -> VALU read-after-write hazard: the ISA of the BUILT libgpamd.so is audited on the CPU.

The full-chip sweep (tests/test_gpu_kv.py) can only see a hazard that happens to strike on the day; this test reads the shipped code
objects instead (scripts/isa_hazard_audit.py) and measures, for every matrix-pipe instruction that writes VGPRs, how many wait states pass
before the first non-matrix instruction reads one of them:
  * kernels that consume MFMA results on the VALU behind ``mfma_result_fence(regs...)`` (kv_gramv / kv_gram4 / kv_gram16 / kv_grad2):
    at least the fence's 32 wait states -- i.e. no MFMA has been moved below its fence by the optimiser (which round 3's fence, lacking
    data dependencies, allowed: two MFMAs of kv_grad2_kernel<.., WSPLIT> were emitted after the s_nops);
  * kv_gram_kernel (the headline fp32-contraction kernel): at least 20 -- 8 explicit wait states behind its Gram MFMAs on top of the table (round 5);
  * kv_gramh_kernel (the library-default split kernel): at least 20 as well since round 6 (8 tied wait states behind every Gram MFMA + one intervening
    contraction MFMA); still stress-tested on the device against its fully fenced build (tests/test_gpu_hazard_stress.py);
  * every other kernel: never closer than the toolchain's own hazard table for an 8-pass XDL write (12 wait states).
A compiler or flag change that breaks either shows up here, at build time, without a GPU."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "scripts"))
LIB = os.path.join(HERE, "..", "gpytorch_amd", "csrc", "libgpamd.so")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libgpamd.so not built")
def test_every_vgpr_destination_mfma_is_read_behind_its_fence():
    import isa_hazard_audit as A

    if not os.path.exists(A.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    rows = A.audit_library(LIB)
    fams = {A.family(r["kernel"]) for r in rows}
    assert set(A.FENCED) <= fams, fams                       # the audit saw the kernels it is meant to guard
    assert len(rows) > 1000                                  # ... in all their instantiations
    # only the Gram MFMA (32x32x16 f16, VGPR destination under the register cap) of kv_gram_kernel carries the raised bar; its contraction MFMAs
    # accumulate in AGPRs / are read in the epilogue behind the zero-argument fence
    def bar(r):
        fam = A.family(r["kernel"])
        if fam == "kv_gram_kernel" and "32x32x16_f16" not in r["mfma"]:
            return A.COMPILER_TABLE
        return A.required(fam)

    bad = [r for r in rows if r["min_wait_states"] < bar(r)]
    assert not bad, bad[:5]
    gram = [r["min_wait_states"] for r in rows if A.family(r["kernel"]) == "kv_gram_kernel" and "32x32x16_f16" in r["mfma"]]
    assert gram and min(gram) >= 20, min(gram)
    gramh = [r["min_wait_states"] for r in rows if A.family(r["kernel"]) == "kv_gramh_kernel"]
    assert gramh and min(gramh) >= 20, min(gramh)


def test_audit_measures_a_synthetic_hazard():
    """The parser / distance rule on a hand-written listing: an MFMA read after s_nop 3 -> 4 wait states; through a back-edge the loop head counts."""
    import isa_hazard_audit as A

    listing = """
	v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], 0       // 000000001000: AAAA
	s_nop 3                                                    // 000000001008: BBBB
	v_exp_f32_e32 v40, v3                                      // 00000000100C: CCCC
	v_mfma_f32_4x4x1_16b_f32 v[24:27], v30, v31, v[24:27]      // 000000001010: DDDD
	s_nop 1                                                    // 000000001018: EEEE
	s_cbranch_scc1 65530                                       // 00000000101C: FFFF <k+0xc>
	s_endpgm                                                   // 000000001020: 0000
"""
    ins = A.parse_function(listing.splitlines())
    res = A.audit_function("k", ins)
    assert res["v_mfma_f32_32x32x16_f16"][0] == 4
    # 4x4x1 result v[24:27] is never read by a non-MFMA instruction: reported as "far" (the horizon)
    assert res["v_mfma_f32_4x4x1_16b_f32"][0] >= 90
    listing2 = listing.replace("v_exp_f32_e32 v40, v3 ", "v_exp_f32_e32 v40, v25")
    res2 = A.audit_function("k", A.parse_function(listing2.splitlines()))
    assert res2["v_mfma_f32_4x4x1_16b_f32"][0] == 3          # s_nop 1 (2) + branch (1), then the loop head reads v25
