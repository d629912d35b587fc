"""Big-int model of the THREE-digit form of the balanced-31-bit-limb dot product (round 4; l31_dot3 in
sda_amd/csrc/sda_kernels.hip): groups of up to SEVEN terms in four signed 64-bit columns (the two cross columns kept apart),
a carry normalisation between groups instead of a reduction, and ONE Montgomery reduction with R = 2^93 (three radix-2^31
digits) at the end - the larger R absorbs the magnitude that makes a two-digit reduction of more than five terms overflow a
64-bit register (its result lies in (-p/2 - eps, p/2 + eps) for any term count).  Every register is checked against its
width, on random and adversarial operands, for the largest primes below 2^62 and for small ones."""
import random

import pytest

B = 1 << 31
MB = B - 1
R93 = 1 << 93


def sext31(x):
    x &= MB
    return x - B if x >= (1 << 30) else x


def bal(x):
    x0 = sext31(x)
    x1 = (x - x0) >> 31
    assert x1 * B + x0 == x and -(1 << 30) <= x1 <= (1 << 30)
    return x0, x1


def i64(x):
    assert -(1 << 63) <= x < (1 << 63), x
    return x


def u32(x):
    assert 0 <= x < (1 << 32), x
    return x


def centre(v, p):
    return v - p if v >= (p + 1) // 2 else v


def host_admits_wide(cons):
    """sda_capi.cpp l31_wide_group_ok (round 5): 9 .. 12 terms as ONE group when, for the ACTUAL constants, no column can leave a
    signed 64-bit register whatever the values (limbs of magnitude <= 2^30): sum |limb| 2^30 + 2^33 < 2^63 for the m0 and the m1
    limbs of the row"""
    if not 9 <= len(cons) <= 12:
        return False
    return (sum(abs(m0) for m0, _ in cons) << 30) + (1 << 33) < (1 << 63) and (sum(abs(m1) for _, m1 in cons) << 30) + (1 << 33) < (1 << 63)


def groups_rt(kt):
    """l31_dot_rt3 (round 5, the run-time (k, t) kernels from nine terms on): groups of seven, the remainder last, whatever its
    size - no 8-term joining, no one-group form; C3 starts at zero and every reduction is the SPLIT0 / HAS_C3 one"""
    return [7] * (kt // 7) + ([kt % 7] if kt % 7 else [])


def groups_of(kt, cons=None):
    """l31_dot3: groups of seven, the remainder last; a remainder of ONE term joins the group before it (7 + 8 for 15 terms),
    which the host admits only after checking the actual constants (host_admits_eight).  Round 5: 9 .. 12 terms form ONE group
    where the constants admit it (host_admits_wide; in the library the answer covers every row of both share maps)"""
    if cons is not None and host_admits_wide(cons):
        return [kt]
    sizes = [7] * (kt // 7) + ([kt % 7] if kt % 7 else [])
    if len(sizes) >= 2 and sizes[-1] == 1:
        sizes = sizes[:-2] + [8]
    return sizes


def host_admits_eight(cons, sizes):
    """sda_capi.cpp: an 8-term last group needs every column to stay inside a signed 64-bit register for ANY values (limbs of
    magnitude <= 2^30): sum |m0| 2^30 + 2^32 < 2^63 and the same for m1 - false only when all eight limbs are -2^30"""
    if sizes[-1] != 8:
        return True
    grp = cons[-8:]
    return (sum(abs(m0) for m0, _ in grp) << 30) + (1 << 32) < (1 << 63) and (sum(abs(m1) for _, m1 in grp) << 30) + (1 << 33) < (1 << 63)


def dot3(p, row, vals, wide=True, seen=None, runtime=False):
    """row: constants m (canonical, NOT yet in Montgomery form); vals: canonical values.  Returns sum m v mod p as the kernel computes it."""
    pinvB = (-pow(p, -1, B)) % B
    p0, p1 = p % B, p >> 31
    cons = [bal(centre(m * R93 % p, p)) for m in row]
    lim = [bal(centre(v, p)) for v in vals]
    C0 = C1a = C1b = C2 = C3 = 0
    g = 0
    sizes = groups_rt(len(row)) if runtime else groups_of(len(row), cons if wide else None)
    if seen is not None:
        seen.append(tuple(sizes))
    if not host_admits_eight(cons, sizes):
        return sum(m * v for m, v in zip(row, vals)) % p          # the library then serves the shape with another kernel
    for gi, size in enumerate(sizes):
        if gi > 0:                                            # ---- l31_normalize: carries forward, no reduction
            t0 = C0 >> 31
            C0 = C0 & MB
            C1a = i64(C1a + t0)
            hi = i64((C1a >> 31) + (C1b >> 31))
            lows = u32((C1a & MB) + (C1b & MB))
            C1a, C1b = lows, 0
            C2 = i64(C2 + hi)
            t2 = C2 >> 31
            C2 = C2 & MB
            C3 = i64(C3 + t2)
        for (m0, m1), (v0, v1) in zip(cons[g:g + size], lim[g:g + size]):
            C0 = i64(C0 + m0 * v0)
            C1a = i64(C1a + m0 * v1)
            C1b = i64(C1b + m1 * v0)
            C2 = i64(C2 + m1 * v1)
        g += size
    X = C0 + (C1a + C1b) * B + C2 * B * B + C3 * B * B * B
    # ---- l31_redc3
    q0 = sext31((C0 & 0xFFFFFFFF) * pinvB)
    if sizes[-1] >= 7 or runtime:
        # a full last group: C0 + q0 p0 can pass 2^63 (7 products of 2^60 + 2^61) - the quotient by B is formed from the floor
        # of C0 and the exact quotient of its low limb plus q0 p0
        low = i64((C0 & MB) + q0 * p0)
        assert low % B == 0
        d0 = i64((C0 >> 31) + (low >> 31))
        assert d0 * B == C0 + q0 * p0
    else:
        D0 = i64(C0 + q0 * p0)
        assert D0 % B == 0
        d0 = D0 >> 31
    E1 = i64(d0 + q0 * p1)
    lows = u32((C1a & MB) + (C1b & MB))
    q1 = sext31(((C1a & 0xFFFFFFFF) + (C1b & 0xFFFFFFFF) + (E1 & 0xFFFFFFFF)) * pinvB)
    F = i64(E1 + q1 * p0)
    assert (C1a + C1b + F) % B == 0
    carry1 = i64((C1a >> 31) + (C1b >> 31) + (i64(F + lows) >> 31))
    assert carry1 == (C1a + C1b + F) // B
    c2lo, c2hi = C2 & MB, C2 >> 31
    G = i64(c2lo + q1 * p1 + carry1)
    q2 = sext31((G & 0xFFFFFFFF) * pinvB)
    H = i64(G + q2 * p0)
    assert H % B == 0
    res = i64(c2hi + (H >> 31) + q2 * p1 + C3)
    assert res * R93 == X + (q0 + q1 * B + q2 * B * B) * p
    assert -p < res < p, (res, p)
    lifted = (res + p) % (1 << 64)                             # unsigned wrap: exactly when res < 0
    out = lifted if lifted < (res % (1 << 64)) else res % (1 << 64)
    assert 0 <= out < p
    return out


def largest_prime_below(n):
    from sympy import prevprime
    return prevprime(n)


PRIMES = [4611686006577364993, None, 433, 746497, (1 << 61) + 1 - 0]      # None -> the largest prime below 2^62


@pytest.mark.parametrize("p", PRIMES)
@pytest.mark.parametrize("kt", [6, 7, 9, 10, 13, 14, 15, 16])
def test_three_digit_dot_product_exact_and_in_range(p, kt):
    if p is None:
        p = largest_prime_below(1 << 62)
    if p == (1 << 61) + 1:
        from sympy import nextprime
        p = nextprime(1 << 61)
    rnd = random.Random(kt * 1000 + p % 997)
    ext = [0, 1, p - 1, (p - 1) // 2, (p + 1) // 2, 1 << 30, (1 << 30) - 1, (p - 1) // 2 - (1 << 30), ((p + 1) // 2 + (1 << 30)) % p]
    # operands whose balanced limbs are extreme: limbs of +-2^30 in both positions
    lim_ext = []
    for a in (-(1 << 30), (1 << 30) - 1):
        for b in (-(1 << 30), (1 << 30) - 1, 1 << 30):
            x = b * B + a
            if abs(x) <= (p - 1) // 2:
                lim_ext.append(x % p)
    for trial in range(300):
        if trial < 20:
            row = [ext[(trial + i) % len(ext)] for i in range(kt)]
            vals = [ext[(trial * 3 + 2 * i) % len(ext)] for i in range(kt)]
        elif trial < 60 and lim_ext:
            # adversarial: constants whose Montgomery form has extreme limbs, values with extreme limbs, all the same sign
            inv = pow(R93, -1, p)
            row = [lim_ext[(trial + i) % len(lim_ext)] * inv % p for i in range(kt)] if trial & 1 else [lim_ext[trial % len(lim_ext)] * inv % p] * kt
            vals = [lim_ext[(trial // 2 + i) % len(lim_ext)] for i in range(kt)] if trial & 2 else [lim_ext[(trial // 3) % len(lim_ext)]] * kt
        else:
            row = [rnd.randrange(p) for _ in range(kt)]
            vals = [rnd.randrange(p) for _ in range(kt)]
        assert dot3(p, row, vals) == sum(m * v for m, v in zip(row, vals)) % p
        assert dot3(p, row, vals, wide=False) == sum(m * v for m, v in zip(row, vals)) % p


@pytest.mark.parametrize("kt", [9, 10, 11, 12])
def test_wide_group_is_taken_when_the_constants_admit_it_and_never_overflows(kt):
    """the one-group form of round 5: random constants admit it most of the time for 9 / 10 terms (sum |limb| ~ kt / 2 x 2^30
    against the 8 x 2^30 a column can take), adversarial ones never - and whenever it is taken every register of the model
    stays inside 64 bits for the WORST values (all limbs +-2^30, signs aligned with the constants)"""
    p = 4611686006577364993
    rnd = random.Random(kt)
    taken = 0
    for trial in range(400):
        row = [rnd.randrange(p) for _ in range(kt)]
        cons = [bal(centre(m * R93 % p, p)) for m in row]
        seen = []
        # worst-case values for THIS row: every limb of magnitude 2^30 (or as close as a residue allows), signs matched to m0
        worst = []
        for m0, m1 in cons:
            s = 1 if m0 >= 0 else -1
            x = s * ((1 << 30) - 1) * B + s * ((1 << 30) - 1)
            worst.append(x % p)
        for vals in (worst, [rnd.randrange(p) for _ in range(kt)]):
            assert dot3(p, row, vals, seen=seen) == sum(m * v for m, v in zip(row, vals)) % p     # i64() asserts inside
        taken += seen[0] == (kt,)
    assert taken > (300 if kt <= 10 else 0), taken
    # BASELINE config 4's own constants (k=8, t=2, n=26 over the 62-bit prime, roots 5^((p-1)/16), 5^((p-1)/27)): admitted
    if kt == 10:
        w2, w3 = 2589100645267092065, 365137883145458390
        nodes = [pow(w2, e, p) for e in range(11)]
        for j in range(26):
            x = pow(w3, j + 1, p)
            row = []
            for a, xa in enumerate(nodes):
                num = den = 1
                for b, xb in enumerate(nodes):
                    if a != b:
                        num = num * (x - xb) % p
                        den = den * (xa - xb) % p
                row.append(num * pow(den, p - 2, p) % p)
            assert host_admits_wide([bal(centre(m * R93 % p, p)) for m in row[1:]]), j


@pytest.mark.parametrize("kt", [9, 12, 13, 16, 17, 31, 32, 33, 48, 63, 64])
def test_run_time_three_digit_form_up_to_64_terms(kt):
    """the run-time (k, t) kernels' grouping (sevens + remainder, up to ten groups): exact and inside every register for random,
    extreme and sign-aligned operands over the 62-bit prime, the largest prime below 2^62 and a small one"""
    for p in (4611686006577364993, largest_prime_below(1 << 62), 746497):
        rnd = random.Random(kt + p % 1000)
        inv = pow(R93, -1, p)
        big = [x % p for x in ((1 << 30) * B + (1 << 30) - 1, -(1 << 30) * B - (1 << 30), ((1 << 30) - 1) * B - (1 << 30)) if abs(x) <= (p - 1) // 2]
        for trial in range(120):
            if trial < 30 and big:
                row = [big[(trial + i) % len(big)] * inv % p for i in range(kt)] if trial & 1 else [big[trial % len(big)] * inv % p] * kt
                vals = [big[(trial // 2 + i) % len(big)] for i in range(kt)] if trial & 2 else [big[(trial // 3) % len(big)]] * kt
            else:
                row = [rnd.randrange(p) for _ in range(kt)]
                vals = [rnd.randrange(p) for _ in range(kt)]
            assert dot3(p, row, vals, runtime=True) == sum(m * v for m, v in zip(row, vals)) % p


def wrap64(x):
    """a signed 64-bit register after wrap-around arithmetic (v_mad_i64_i32 and 64-bit adds / subs wrap modulo 2^64)"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def i32(x):
    assert -(1 << 31) <= x < (1 << 31), x
    return x


def host_admits_karatsuba(cons):
    """sda_capi.cpp l31_karatsuba_ok: the wide group is admitted AND each half's cross column fits a signed 64-bit register for any
    values (limbs of magnitude <= 2^30): sum over the half of (|m0| + |m1|) 2^30 + 2^33 < 2^63"""
    if not host_admits_wide(cons):
        return False
    half = (len(cons) + 1) // 2
    return all((sum(abs(m0) + abs(m1) for m0, m1 in cons[lo:lo + half]) << 30) + (1 << 33) < (1 << 63) for lo in range(0, len(cons), half))


def karatsuba_columns(cons, lim):
    """l31_dot3_wide_k (round 6): two halves, each with C0, C2 and a middle column M = sum (m0 + m1)(v0 + v1) in WRAP-AROUND
    arithmetic; the halves' cross columns M - C0 - C2 take the places of C1a and C1b.  Returns (C0, C1a, C1b, C2) and asserts
    every operand / result the kernel relies on: the 32-bit sums, the halves' cross columns, the summed C0 and C2."""
    kt = len(cons)
    half = (kt + 1) // 2
    parts = []
    for lo, hi in ((0, half), (half, kt)):
        C0 = C2 = M = 0
        for (m0, m1), (v0, v1) in zip(cons[lo:hi], lim[lo:hi]):
            ms, vs = i32(m0 + m1), i32(v0 + v1)                  # both fit a signed 32-bit register
            C0 = wrap64(C0 + m0 * v0)
            C2 = wrap64(C2 + m1 * v1)
            M = wrap64(M + ms * vs)                              # may have wrapped: only M - C0 - C2 is used
        cross = wrap64(M - C0 - C2)
        exact = sum(m0 * v1 + m1 * v0 for (m0, m1), (v0, v1) in zip(cons[lo:hi], lim[lo:hi]))
        assert cross == exact and -(1 << 63) <= exact < (1 << 63), (cross, exact)
        parts.append((C0, C2, cross, sum(m0 * v0 for (m0, _), (v0, _) in zip(cons[lo:hi], lim[lo:hi])),
                      sum(m1 * v1 for (_, m1), (_, v1) in zip(cons[lo:hi], lim[lo:hi]))))
    C0 = i64(parts[0][3] + parts[1][3])                          # the host's check keeps the WHOLE columns inside 64 bits,
    C2 = i64(parts[0][4] + parts[1][4])                          # so the halves never wrapped either
    assert parts[0][0] == parts[0][3] and parts[1][0] == parts[1][3] and parts[0][1] == parts[0][4] and parts[1][1] == parts[1][4]
    return C0, parts[0][2], parts[1][2], C2


@pytest.mark.parametrize("kt", [9, 10, 11, 12])
def test_karatsuba_halves_give_the_plain_columns(kt):
    """the Karatsuba form of the wide group: for constants the host admits, ANY values (random, and the worst ones: every limb at
    its extreme, signs aligned with the constants so that the middle column wraps as far as it can) give C0, C2 and a pair of
    cross columns whose SUM is the plain form's C1a + C1b - which is all l31_redc3 uses - with every half's cross column inside
    a signed 64-bit register - for the constants the host ADMITS (host_admits_karatsuba); constants it refuses do overflow on the
    extreme values, which the last lines show"""
    p = 4611686006577364993
    rnd = random.Random(100 + kt)
    checked = wrapped = 0
    refused = []
    for trial in range(400):
        row = [rnd.randrange(p) for _ in range(kt)]
        cons = [bal(centre(m * R93 % p, p)) for m in row]
        if not host_admits_karatsuba(cons):
            refused.append(cons)
            continue
        extremes = []
        for m0, m1 in cons:
            s0, s1 = (1 if m0 >= 0 else -1), (1 if m1 >= 0 else -1)
            extremes.append(((1 << 30) - 1 if s0 > 0 else -(1 << 30), (1 << 30) if s1 > 0 else -(1 << 30)))
        rand_lim = [bal(centre(rnd.randrange(p), p)) for _ in range(kt)]
        for lim in (extremes, rand_lim, [(-(1 << 30), -(1 << 30))] * kt, [((1 << 30) - 1, 1 << 30)] * kt):
            C0, C1a, C1b, C2 = karatsuba_columns(cons, lim)
            assert C0 == sum(m0 * v0 for (m0, _), (v0, _) in zip(cons, lim))
            assert C2 == sum(m1 * v1 for (_, m1), (_, v1) in zip(cons, lim))
            assert C1a + C1b == sum(m0 * v1 + m1 * v0 for (m0, m1), (v0, v1) in zip(cons, lim))
            mid = sum((m0 + m1) * (v0 + v1) for (m0, m1), (v0, v1) in zip(cons[:(kt + 1) // 2], lim[:(kt + 1) // 2]))
            wrapped += not -(1 << 63) <= mid < (1 << 63)
            checked += 1
    assert checked > (400 if kt <= 10 else 0) and (wrapped > 0 or checked == 0), (checked, wrapped)   # the middle column DID wrap: the identity held anyway
    # constants with every limb at its extreme are refused, and rightly: their halves' cross columns leave the register
    bad = [((1 << 30) - 1, 1 << 30)] * kt
    assert not host_admits_karatsuba(bad)
    with pytest.raises(AssertionError):
        karatsuba_columns(bad, [((1 << 30) - 1, 1 << 30)] * kt)
    # BASELINE config 4's own constants (tss's map): admitted
    if kt == 10:
        w2, w3 = 2589100645267092065, 365137883145458390
        nodes = [pow(w2, e, p) for e in range(11)]
        for j in range(26):
            x = pow(w3, j + 1, p)
            row = []
            for a, xa in enumerate(nodes):
                num = den = 1
                for b, xb in enumerate(nodes):
                    if a != b:
                        num = num * (x - xb) % p
                        den = den * (xa - xb) % p
                row.append(num * pow(den, p - 2, p) % p)
            assert host_admits_karatsuba([bal(centre(m * R93 % p, p)) for m in row[1:]]), j
