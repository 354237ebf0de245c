"""The oracle's AVX512-FP16 tier (oracle/vso.c f16_fp16acc: fp16 rows of dim >= 32 accumulated in HALF precision, what a reference
built by gcc >= 12 runs on avx512_fp16 hosts: IP_AVX512FP16_VL_FP16.h:16-51, L2_AVX512FP16_VL_FP16.h:16-58).  Parity for this tier is
UNPINNED -- nothing in this image or on the GPU box can execute those kernels -- so what can be checked is checked: the half
arithmetic against an independent exact implementation, the kernel's shape against a plain restatement in numpy float16, and the
property the reference's own test asserts (test_spaces.cpp:1418-1590: within 1 % of a sequential half-precision sum)."""
from fractions import Fraction

import numpy as np
import pytest

from util import TIERS, TYPES, random_vectors


def _val(h):
    return Fraction(float(np.array([h], np.uint16).view(np.float16)[0]))


def _round_half(x):
    """Fraction -> IEEE half bits, nearest-even; None for an exact zero (its sign is the operation's business)"""
    if x == 0:
        return None
    neg, x = x < 0, abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    while Fraction(2) ** e > x:
        e -= 1
    while Fraction(2) ** (e + 1) <= x:
        e += 1
    qe = -24 if e < -14 else e - 10
    m = x / Fraction(2) ** qe
    f = m.numerator // m.denominator
    rem = m - f
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and f & 1):
        f += 1
    v = Fraction(f) * Fraction(2) ** qe
    bits = 0x7C00 if v >= 65520 else int(np.array([float(v)], np.float64).astype(np.float16).view(np.uint16)[0])
    return bits | (0x8000 if neg else 0)


def _finite_halves(rng, n):
    h = rng.integers(0, 65536, n).astype(np.uint16)
    return h[(h & 0x7C00) != 0x7C00]


def test_half_fma_is_one_exact_rounding(vso):
    rng = np.random.default_rng(5)
    a, b, c = (_finite_halves(rng, 6000) for _ in range(3))
    n = min(len(a), len(b), len(c))
    for i in range(n):
        x, y, z = int(a[i]), int(b[i]), int(c[i])
        if i % 3 == 0:   # an addend close to -x y: cancellation, ties, results in the subnormal range
            with np.errstate(over="ignore"):
                z2 = int(np.array([float(-(_val(x) * _val(y)))], np.float64).astype(np.float16).view(np.uint16)[0]) ^ int(rng.integers(0, 4))
            if (z2 & 0x7C00) != 0x7C00:
                z = z2
        want = _round_half(_val(x) * _val(y) + _val(z))
        got = vso.h_fma(x, y, z)
        assert (got & 0x7FFF) == 0 if want is None else got == want, (hex(x), hex(y), hex(z), hex(got), want)
    # specials and the ends of the range
    INF, NINF, ONE, MAX, TINY = 0x7C00, 0xFC00, 0x3C00, 0x7BFF, 0x0001
    assert vso.h_fma(MAX, MAX, 0) == INF and vso.h_fma(MAX, 0xFBFF, 0) == NINF
    assert vso.h_fma(INF, ONE, NINF) & 0x7FFF > 0x7C00          # inf - inf: NaN
    assert vso.h_fma(INF, 0, ONE) & 0x7FFF > 0x7C00             # inf x 0: NaN
    assert vso.h_fma(TINY, TINY, 0) == 0 and vso.h_fma(TINY, 0x8001, 0) == 0x8000   # underflow keeps the sign
    assert vso.h_fma(TINY, ONE, TINY) == 0x0002                 # subnormals are kept, not flushed
    assert vso.h_fma(0x8000, ONE, 0x8000) == 0x8000 and vso.h_fma(0x8000, ONE, 0) == 0


def test_half_mul_add_sub_equal_numpy_float16(vso):
    rng = np.random.default_rng(6)
    a, b = _finite_halves(rng, 30000), _finite_halves(rng, 30000)
    n = min(len(a), len(b))
    a, b = a[:n], b[:n]
    fa, fb = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
    with np.errstate(all="ignore"):   # (exact in double, then ONE rounding to half)
        wm, wa, ws = ((fa * fb).astype(np.float16).view(np.uint16), (fa + fb).astype(np.float16).view(np.uint16),
                      (fa - fb).astype(np.float16).view(np.uint16))
    for i in range(n):
        x, y = int(a[i]), int(b[i])
        assert vso.h_mul(x, y) == int(wm[i]) and vso.h_add(x, y) == int(wa[i]) and vso.h_sub(x, y) == int(ws[i]), (hex(x), hex(y))


def _restated(a, b, l2):
    """the kernel's shape in numpy float16 (element-wise float16 operations round once; the fma through exact Fractions)"""
    d = len(a)
    s = [0] * 32
    ab, bb = a.view(np.uint16), b.view(np.uint16)
    res = d % 32
    sub = lambda x, y: int((np.float16(np.array([x], np.uint16).view(np.float16)[0]) - np.array([y], np.uint16).view(np.float16)[0]).view(np.uint16))  # noqa: E731

    def fma(x, y, z):
        r = _round_half(_val(x) * _val(y) + _val(z))
        return 0 if r is None else r
    for j in range(res):
        x, y = int(ab[j]), int(bb[j])
        if l2:
            x = y = sub(x, y)
        s[j] = fma(x, y, 0)
    for pos in range(res, d, 32):
        for j in range(32):
            x, y = int(ab[pos + j]), int(bb[pos + j])
            if l2:
                x = y = sub(x, y)
            s[j] = fma(x, y, s[j])
    v = np.array(s, np.uint16).view(np.float16)
    for o in (16, 8, 4, 2, 1):
        v = (v[:o] + v[o:2 * o]).astype(np.float16)
    r = v[0] if l2 else np.float16(1) - v[0]
    return float(np.float16(r))


@pytest.mark.parametrize("dim", [32, 33, 47, 63, 64, 65, 100, 257])
def test_fp16_tier_kernel_shape_and_the_references_own_tolerance(vso, dim):
    rng = np.random.default_rng(dim)
    a = rng.uniform(-0.99, 0.99, dim).astype(np.float16)
    b = rng.uniform(-0.99, 0.99, dim).astype(np.float16)
    for metric, l2 in (("L2", True), ("IP", False)):
        got = vso.distance(TYPES["f16"], {"L2": vso.L2, "IP": vso.IP}[metric], a, b, dim, tier=TIERS["avx512_fp16"])
        assert got == _restated(a, b, l2), (metric, dim)
        # the reference's own assertion for this tier: within 1 % of the sequential half-precision baseline
        base = np.float16(0)
        for i in range(dim):
            t = (a[i] - b[i]) if l2 else None
            base = np.float16(base + (np.float16(t * t) if l2 else np.float16(a[i] * b[i])))
        base = float(base if l2 else np.float16(1) - base)
        assert abs(got / base - 1) <= 0.01, (metric, dim, got, base)
    # below 32 elements the tier has no kernel of its own: the AVX512F / F16C order
    for d in (8, 15, 16, 31):
        x, y = rng.uniform(-1, 1, d).astype(np.float16), rng.uniform(-1, 1, d).astype(np.float16)
        assert vso.distance(TYPES["f16"], vso.IP, x, y, d, tier=TIERS["avx512_fp16"]) == vso.distance(TYPES["f16"], vso.IP, x, y, d, tier=TIERS["avx512"])
    # bf16 IP: the vdpbf16ps kernel (every avx512_fp16 CPU has avx512_bf16)
    x, y = random_vectors(rng, 2, 96, "bf16", vso)
    assert vso.distance(TYPES["bf16"], vso.IP, x, y, 96, tier=TIERS["avx512_fp16"]) == vso.distance(TYPES["bf16"], vso.IP, x, y, 96, tier=TIERS["avx512_bf16"])
