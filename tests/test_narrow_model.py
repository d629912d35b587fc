"""Big-int model of the narrow-modulus arithmetic (sda_amd/csrc/narrow_gen.inc.hpp): one signed 32-bit limb per residue,
GROUP terms per signed 64-bit sum, a three-instruction Montgomery reduction with R = 2^32.  Every register is checked
against its width, every bound stated in the kernel's comments is asserted, and the result is compared with plain
modular arithmetic - for random and extreme operands, for primes from 3 to just below 2^31."""
import random

import pytest


def s32(x):
    assert -(1 << 31) <= x < (1 << 31), x
    return x


def wrap32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def centre(v, p):
    h = (p + 1) // 2
    return s32(v - p if v >= h else v)


def group_for(p):
    return 16 if p < (1 << 29) else 4          # launchers: GROUP * p < 2^33


def redc(S, p, pinv):
    """n31_redc: S -> S * 2^-32 mod p in [0, p)"""
    assert -(1 << 63) <= S < (1 << 63)
    sl = S & 0xFFFFFFFF
    sh = s32(S >> 32)
    q = wrap32(sl * pinv)
    assert (S + q * p) % (1 << 32) == 0
    mulhi = (q * p) >> 32                                   # v_mul_hi_i32 (floor)
    t = sh + s32(mulhi) + (1 if sl else 0)
    assert t == (S + q * p) >> 32 and -p < t < p, (t, p)
    s32(t)
    t += p if t < 0 else 0
    assert 0 <= t < p and (t * (1 << 32) - S) % p == 0
    return t


def dot(row, vals, p, pinv, group):
    acc = None
    for g0 in range(0, len(row), group):
        S = 0
        for m, v in zip(row[g0:g0 + group], vals[g0:g0 + group]):
            S += s32(m) * s32(v)
            assert -(1 << 62) < S < (1 << 62)
        r = redc(S, p, pinv)
        if acc is None:
            acc = r
        else:
            s = acc + r
            assert s < (1 << 32)
            d = (s - p) & 0xFFFFFFFF
            acc = d if d < s else s
            assert 0 <= acc < p
    return acc


PRIMES = [3, 433, 746497, 5038849, (1 << 29) - 3, 536870923, (1 << 30) + 3, 2147483629, 2147483647]


@pytest.mark.parametrize("p", PRIMES)
def test_narrow_dot_products_exact(p):
    assert all(p % d for d in range(2, min(p, 50000)) if d * d <= p)          # small trial division (the list is fixed)
    rnd = random.Random(p)
    pinv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    R = (1 << 32) % p
    group = group_for(p)
    assert group * p < (1 << 33)
    extremes = [0, 1, p - 1, (p - 1) // 2, (p + 1) // 2, (p + 1) // 2 - 1 if p > 3 else 0]
    for kt in (1, 4, 5, 7, 10, 15, 16):
        for trial in range(60):
            if trial < 6:
                M = [extremes[(trial + i) % len(extremes)] for i in range(kt)]
                V = [extremes[(trial + 2 * i) % len(extremes)] for i in range(kt)]
            elif trial < 12:                                                       # worst-case magnitudes, same signs
                M = [(p - 1) // 2 if trial & 1 else (p + 1) // 2] * kt
                V = [(p - 1) // 2 if trial & 2 else (p + 1) // 2] * kt
            else:
                M = [rnd.randrange(p) for _ in range(kt)]
                V = [rnd.randrange(p) for _ in range(kt)]
            row = [centre(m * R % p, p) for m in M]                                # Montgomery-form constants, centred
            vals = [centre(v, p) for v in V]
            want = sum(m * v for m, v in zip(M, V)) % p
            assert dot(row, vals, p, pinv, group) == want


@pytest.mark.parametrize("p", [433, 746497, 5038849, (1 << 29) - 3, 1073738161, (1 << 30) - 35])
def test_narrow_shoup_product_and_butterfly_ranges(p):
    """the uint32_t instantiation of the transform kernel (fft_kernels.hip): Shoup product x w - hi32(x ws) p in the low 32
    bits lies in [0, 2p) for ANY 32-bit x; the lazy ranges [0, 4p) of the radix-3 butterfly fit 32 bits when p < 2^30."""
    assert p < (1 << 30) and 4 * p <= (1 << 32)
    rnd = random.Random(p)
    M = 1 << 32
    for trial in range(3000):
        w = rnd.randrange(p) if trial > 5 else [0, 1, p - 1, p // 2, 2, p - 2][trial]
        ws = (w << 32) // p
        assert ws < M
        x = rnd.randrange(M) if trial % 7 else [0, M - 1, 4 * p - 1, 2 * p, p][trial % 5]
        q = (x * ws) >> 32
        r = (x * w - q * p) % M
        assert r == x * w - q * p and 0 <= r < 2 * p and r % p == x * w % p
    def csub(x, m):
        d = (x - m) % M
        return d if d < x else x
    p2 = 2 * p
    for trial in range(3000):
        A, B, C = (rnd.randrange(p2) for _ in range(3))
        om = rnd.randrange(p)
        oms = (om << 32) // p
        t = B + p2 - C
        assert t < M
        qq = (t * oms) >> 32
        w = (t * om - qq * p) % M
        assert w < p2
        y0 = csub(A + B, p2) + C
        y1 = csub(A + p2 - C, p2) + w
        y2 = csub(A + p2 - B, p2) + (p2 - w)
        for y in (y0, y1, y2):
            assert 0 <= y < 4 * p <= M
        assert y0 % p == (A + B + C) % p and y1 % p == (A - C + om * (B - C)) % p and y2 % p == (A - B - om * (B - C)) % p


@pytest.mark.parametrize("p,b", [(433, 2), (746497, 6), (5038849, 9), ((1 << 26) - 5, 6), (104857601, 9)])
def test_lazy_radix3_levels_never_wrap(p, b):
    """LAZY mode of the narrow transform kernel (fft_kernels.hip, (4 b + 4) p < 2^32): no conditional subtraction in any of
    the b radix-3 levels; the A chain grows by at most 4p per level from 2p, B and C always come out of a Shoup product
    (any 32-bit operand -> [0, 2p)), the last pass reduces once with the companion of 1.  Worst-case and random chains."""
    M = 1 << 32
    assert (4 * b + 4) * p < M
    rnd = random.Random(p + b)
    ones = M // p

    def shoup(x, w):
        assert 0 <= x < M
        ws = (w << 32) // p
        r = (x * w - ((x * ws) >> 32) * p) % M
        assert r == x * w - ((x * ws) >> 32) * p and 0 <= r < 2 * p
        return r

    p2 = 2 * p
    for trial in range(400):
        worst = trial < 8
        A = p2 - 1 if worst else rnd.randrange(p2)
        true = A % p
        bound = p2
        om = rnd.randrange(1, p)
        for level in range(b):
            # B and C are arbitrary earlier values of the chain (< the current bound) times twiddles
            xb, xc = (bound - 1 if worst else rnd.randrange(bound)), (bound - 1 if worst else rnd.randrange(bound))
            wb, wc = rnd.randrange(p), rnd.randrange(p)
            B, C = shoup(xb, wb), shoup(xc, wc)
            if worst:
                B, C = (p2 - 1, 0) if level & 1 else (0, p2 - 1)
                xb, wb, xc, wc = B, 1, C, 1
            w = shoup(B + p2 - C, om)
            y = [A + B + C, A + (p2 - C) + w, A + (p2 - B) + (p2 - w)]
            bound += 4 * p
            assert all(0 <= v < bound <= M for v in y)
            want = [(true + xb * wb + xc * wc) % p, (true - xc * wc + om * (xb * wb - xc * wc)) % p,
                    (true - xb * wb - om * (xb * wb - xc * wc)) % p]
            assert [v % p for v in y] == want
            pick = rnd.randrange(3) if not worst else max(range(3), key=lambda i: y[i])
            A, true = y[pick], want[pick]
        full = (A - ((A * ones) >> 32) * p) % M                       # f_full2: the Shoup product by 1
        assert 0 <= full < p2 and full % p == true
        d = (full - p) % M
        assert (d if d < full else full) == true
