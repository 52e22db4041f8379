"""CPU emulation of the split-f16 Gram block (gpytorch_amd/csrc/gram_f16.hpp): the same slot scheme -- every float32
operand x = hi + lo with hi = f16(x), lo = f16(x - hi); NT = 4 product terms per dimension (3 for D >= 12) plus the split
norms -- evaluated with exact f16 x f16 products and float32 accumulation, as v_mfma_f32_32x32x16_f16 does.  Pins the
accuracy claim of DESIGN.md 3.1 (|dK| <= 2e-5 at max |z|^2 = 32) for every D the kernels are instantiated for, without a
GPU, and the hi/lo consistency requirement that the device code enforces with an asm barrier."""
import pytest
import torch


def _split(x):
    hi = x.to(torch.float16)
    lo = (x - hi.to(torch.float32)).to(torch.float16)
    return hi, lo


def _gram_sq(zj, zi):
    """S[j][i] = |z_j - z_i|^2 from split operands; products of f16 values are exact in float32."""
    d = zj.shape[-1]
    nt = 4 if d <= 11 else 3
    ah, al = _split(zj)
    bh, bl = _split(-2.0 * zi)
    njh, njl = _split((zj * zj).sum(-1))
    nih, nil = _split((zi * zi).sum(-1))
    f = torch.float32
    terms = [ah.to(f) @ bh.to(f).t(), ah.to(f) @ bl.to(f).t(), al.to(f) @ bh.to(f).t()]
    if nt == 4:
        terms.append(al.to(f) @ bl.to(f).t())
    s = sum(terms)
    return s + (njh.to(f) + njl.to(f)).unsqueeze(-1) + (nih.to(f) + nil.to(f)).unsqueeze(0)


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 10, 12, 16])
@pytest.mark.parametrize("zmax2", [1.0, 8.6, 32.0])
def test_split_f16_squared_distances(d, zmax2):
    g = torch.Generator().manual_seed(100 * d + int(zmax2))
    n = 400
    z = torch.rand(2 * n, d, generator=g) * 2 - 1
    z = z * (zmax2 ** 0.5) / z.norm(dim=-1).max()      # max |z|^2 == zmax2 (the host's selection rule bounds this by 32)
    zj, zi = z[:n], z[n:]
    s = _gram_sq(zj, zi)
    ref = (zj.double().unsqueeze(1) - zi.double().unsqueeze(0)).pow(2).sum(-1)
    err = float((s.double() - ref).abs().max())
    # ~2^-22 (|z_i| + |z_j|)^2 from the split (twice that for the three-term form, D >= 12) plus the float32 rounding of S
    # itself (half an ulp of values up to 4 zmax2) -- the latter only where K = 2^-S has long vanished
    assert err < 3e-5 * max(zmax2 / 32.0, 0.05) * (2.0 if d >= 12 else 1.0), (d, zmax2, err)
    k_err = float((torch.exp2(-s.double()) - torch.exp2(-ref)).abs().max())
    assert k_err < 2e-5 * max(zmax2 / 32.0, 0.1), (d, zmax2, k_err)  # the bound the host's |z|^2 <= 32 rule promises


def test_hi_lo_must_come_from_the_same_rounding():
    """A norm one float32 ulp away from an f16 rounding tie: taking hi from one copy and lo from the other is off by a whole
    f16 ulp (what scripts/micro/pack_check.hip caught on the device); the consistent split is exact to 2^-22."""
    nn = torch.tensor(3.9326169490814209)                 # |z|^2 of the point that exposed it
    other = torch.nextafter(nn, torch.tensor(10.0))        # the recomputed copy, 1 ulp up: rounds to the other neighbour
    hi_a, lo_a = _split(nn)
    hi_b, _ = _split(other)
    assert float(hi_a) != float(hi_b)
    consistent = float(hi_a) + float(lo_a)
    mixed = float(hi_b) + float(lo_a)
    assert abs(consistent - float(nn)) < 2 ** -21
    assert abs(mixed - float(nn)) > 1e-3
