"""Big-int model of the narrow limb GEMM (sda_amd/csrc/ngemm_kernels.hip): residues of a prime up to 0x7F7F7F as three balanced
base-256 digits, five signed 32-bit column sums per share (what v_mfma_i32_16x16x64_i8 accumulates), the epilogue's
sum_j C_j c_j with c_j = 256^j 2^32 mod p in a signed 64-bit register and the Montgomery reduction with R = 2^32.  Every
register is checked against its width and the result against plain modular arithmetic, for random and extreme operands at
the largest term count the kernels take (k + t = 512)."""
import random

import pytest

PRIMES = [3, 433, 746497, 5038849, 8355691]          # 8355691: the largest prime the kernels take (p <= 0x7F7F7F)


def digits(v, p, centred):
    """ng_digits (values: the canonical residue itself) / the host's split of the matrix (centred representative): three balanced
    base-256 digits"""
    h = (p + 1) // 2
    c = (v - p if v >= h else v) if centred else v
    x = c & 0xFFFFFFFF
    y = ((x + 0x00808080) & 0xFFFFFFFF) ^ 0x00808080
    d = [((y >> (8 * i)) & 0xFF) for i in range(3)]
    d = [b - 256 if b >= 128 else b for b in d]
    assert d[0] + 256 * d[1] + 65536 * d[2] == c, (v, p, d)
    return d


def wrap32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def share(row, vals, p):
    """one dot product the way the kernel forms it; returns the canonical share"""
    C = [0] * 5
    for m, v in zip(row, vals):
        dm, dv = digits(m, p, True), digits(v, p, False)
        for a in range(3):
            for b in range(3):
                C[a + b] += dm[a] * dv[b]
    for c in C:
        assert -(1 << 31) <= c < (1 << 31)                       # the MFMA accumulators
    cj = []
    x = (1 << 32) % p
    for _ in range(5):
        cj.append(x - p if x > (p - 1) // 2 else x)
        x = x * 256 % p
    # columns 3 and 4 share one multiplication (c_4 = 256 c_3 mod p): C_3 + 256 C_4 in a 32-bit register
    top = C[3] + 256 * C[4]
    assert -(1 << 31) <= top < (1 << 31)
    assert (cj[4] - 256 * cj[3]) % p == 0
    S = 0
    for c, k in zip(C[:3] + [top], cj[:4]):
        S += c * k
        assert -(1 << 63) <= S < (1 << 63)                        # v_mad_i64_i32 chain
    assert abs(S) < p << 31                                       # the reduction's operand bound
    pinv = pow(p, -1, 1 << 32)
    sl, sh = S & 0xFFFFFFFF, S >> 32
    q = wrap32(sl * pinv)
    assert (S - q * p) % (1 << 32) == 0
    t = sh - ((q * p) >> 32)                                      # v_mul_hi_i32 (floor): no borrow, the low words are equal
    assert -(1 << 31) <= sh < (1 << 31)
    assert t == (S - q * p) >> 32 and -p < t < p
    tu = t & 0xFFFFFFFF
    u = (tu + p) & 0xFFFFFFFF
    r = u if u < tu else tu                                       # v_min_u32
    assert 0 <= r < p
    return r


@pytest.mark.parametrize("p", PRIMES)
def test_digits_cover_every_residue_class_edge(p):
    for v in {0, 1, p - 1, p // 2, p // 2 + 1, (p + 1) // 2, max(0, p // 2 - 1), 127, 128, 129, 32767, 32768, 32896} | \
            {random.Random(p).randrange(p) for _ in range(2000)}:
        if 0 <= v < p:
            digits(v, p, True)
            digits(v, p, False)


@pytest.mark.parametrize("p", PRIMES)
@pytest.mark.parametrize("terms", [17, 63, 255, 512])
def test_dot_products_exact(p, terms):
    rng = random.Random(p * 1000 + terms)
    h = p // 2
    extremes = [h, h + 1, p - 1, 0, 1]
    for trial in range(12):
        if trial < 5:                                             # all operands at one extreme: the largest column sums
            row = [extremes[trial]] * terms
            vals = [extremes[(trial * 2) % 5]] * terms
        elif trial < 8:
            row = [rng.choice(extremes) for _ in range(terms)]
            vals = [rng.choice(extremes) for _ in range(terms)]
        else:
            row = [rng.randrange(p) for _ in range(terms)]
            vals = [rng.randrange(p) for _ in range(terms)]
        want = sum(m * v for m, v in zip(row, vals)) % p
        assert share(row, vals, p) == want


def test_column_bound_holds_for_any_operands_at_512_terms():
    # |digit| <= 128, three digit pairs share the middle column: 3 * 512 * 128 * 128 < 2^31
    assert 3 * 512 * 128 * 128 < 1 << 31
    # sum_j |C_j| <= terms * (sum |dM|) (sum |dV|) <= 512 * 384 * 384, |c_j| <= p / 2: |S| < p 2^31
    assert 512 * 384 * 384 // 2 < 1 << 31
    # the merged top column: a centred matrix entry is below 2^22 in magnitude, so its top digit is at most 65 (2^22 + 128 * 257
    # over 65536); a value's top digit at most 128: |C_3 + 256 C_4| <= 512 (128 * 128 + 65 * 128) + 256 * 512 * 65 * 128 < 2^31,
    # and with it |S| <= (p / 2) (2 * 3 * 512 * 2^14 + 2^31 * 0.52) < p 2^31
    top = 512 * (128 * 128 + 65 * 128) + 256 * 512 * 65 * 128
    assert top < 1 << 31 and (3 * 3 * 512 * 128 * 128 + top) // 2 < 1 << 31
    for p in PRIMES:
        h = (p + 1) // 2
        assert max(abs(digits(v, p, True)[2]) for v in (h - 1, h, h + 1, p - 1, 0, 1) if 0 <= v < p) <= 65
