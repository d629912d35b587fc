"""Big-int model of the balanced-31-bit-limb Montgomery dot product used by packed_gen_l31_kernel
(sda_amd/csrc/sda_kernels.hip): checks exactness and that every intermediate fits the signed 64-bit /
32-bit registers the kernel keeps it in, on random and adversarial inputs.  CPU only."""
import random

import pytest

from oracle import pyoracle as po

B = 1 << 31
MB = B - 1


def sext31(x):
    x &= MB
    return x - B if x >= (1 << 30) else x


def bal(x):
    x0 = sext31(x)
    x1 = (x - x0) >> 31
    assert x1 * B + x0 == x
    return x0, x1


def i64(x):
    assert -(1 << 63) <= x < (1 << 63), x
    return x


def i32(x):
    assert -(1 << 31) <= x < (1 << 31), x
    return x


def prep(p, rows):
    """host side: build_l31() in sda_capi.cpp"""
    out = []
    for row in rows:
        r = []
        for m in row:
            mr = (m << 62) % p
            if mr > (p - 1) // 2:
                mr -= p
            r.append(bal(mr))
        out.append(r)
    return dict(p=p, pinvB=(-pow(p, -1, B)) % B, p0=p % B, p1=p >> 31, M=out, h=(p + 1) // 2)


def partition(kt):
    """l31_dot(): groups of four, or of five where that saves a group"""
    g4, g5 = (kt + 3) // 4, (kt + 4) // 5
    if g5 < g4:
        fives = kt - 4 * g5
        return [5] * fives + [4] * (g5 - fives)
    return [4] * (kt // 4) + ([kt % 4] if kt % 4 else [])


def share(ctx, row, v):
    """device side: l31_dot(): groups of <= 4 (or 5) terms, partial results kept lazily in [0, 2p)"""
    p = ctx["p"]
    r = None
    g = 0
    sizes = partition(len(v))
    assert sum(sizes) == len(v)
    for size in sizes:
        top = (group5 if size == 5 else group)(ctx, row[g:g + size], v[g:g + size])
        g += size
        u = top + 2 * p
        assert 0 <= u < 4 * p and u < (1 << 64)
        if u >= 2 * p:
            u -= 2 * p
        if r is None:
            r = u
        else:
            r += u
            assert r < 4 * p and r < (1 << 64)
            if r >= 2 * p:
                r -= 2 * p
    if r >= p:
        r -= p
    assert 0 <= r < p
    return r


def group(ctx, row, v):
    """device side: centre_limbs() + l31_group()"""
    p = ctx["p"]
    C0 = C1 = C2 = 0
    assert len(v) <= 4
    for (m0, m1), x in zip(row, v):
        xc = x - p if x >= ctx["h"] else x
        v0, v1 = bal(xc)
        for w in (v0, v1, m0, m1):
            i32(w)
        assert abs(v1) <= 1 << 30 and abs(m1) <= 1 << 30
        C0 = i64(C0 + m0 * v0); C1 = i64(C1 + m0 * v1); C1 = i64(C1 + m1 * v0); C2 = i64(C2 + m1 * v1)
    q0 = sext31(((C0 & 0xFFFFFFFF) * ctx["pinvB"]) & 0xFFFFFFFF)
    C0 = i64(C0 + q0 * ctx["p0"])
    assert C0 % B == 0
    E = i64((C0 >> 31) + q0 * ctx["p1"])
    q1 = sext31((((C1 & 0xFFFFFFFF) + (E & 0xFFFFFFFF)) * ctx["pinvB"]) & 0xFFFFFFFF)
    E = i64(E + q1 * ctx["p0"])
    assert (C1 + E) % B == 0
    carry1 = (C1 >> 31) + (i64(E + MB) >> 31)
    assert carry1 == (C1 + E) // B
    return i64(i64(C2 + q1 * ctx["p1"]) + carry1)


def group5(ctx, row, v):
    """device side: l31_group5(): the cross columns kept apart, one reduction for five terms"""
    p = ctx["p"]
    C0 = C1a = C1b = C2 = 0
    assert len(v) == 5
    for (m0, m1), x in zip(row, v):
        xc = x - p if x >= ctx["h"] else x
        v0, v1 = bal(xc)
        for w in (v0, v1, m0, m1):
            i32(w)
        C0 = i64(C0 + m0 * v0); C1a = i64(C1a + m0 * v1); C1b = i64(C1b + m1 * v0); C2 = i64(C2 + m1 * v1)
    q0 = sext31(((C0 & 0xFFFFFFFF) * ctx["pinvB"]) & 0xFFFFFFFF)
    C0 = i64(C0 + q0 * ctx["p0"])
    assert C0 % B == 0
    E = i64((C0 >> 31) + q0 * ctx["p1"])
    q1 = sext31((((C1a & 0xFFFFFFFF) + (C1b & 0xFFFFFFFF) + (E & 0xFFFFFFFF)) * ctx["pinvB"]) & 0xFFFFFFFF)
    E = i64(E + q1 * ctx["p0"])
    assert (C1a + C1b + E) % B == 0
    lows = (C1a & MB) + (C1b & MB)
    assert lows < (1 << 32)
    carry1 = (C1a >> 31) + (C1b >> 31) + (i64(E + lows) >> 31)
    assert carry1 == (C1a + C1b + E) // B
    top = i64(i64(C2 + q1 * ctx["p1"]) + carry1)
    assert -1.75 * p < top < 1.75 * p
    return top


def _roots(p, o2, o3):
    g = next(g for g in range(2, 200) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    return pow(g, (p - 1) // o2, p), pow(g, (p - 1) // o3, p)


def test_partition():
    assert partition(15) == [5, 5, 5] and partition(10) == [5, 5] and partition(7) == [4, 3] and partition(9) == [5, 4]
    assert partition(13) == [5, 4, 4] and partition(4) == [4] and partition(5) == [5] and partition(11) == [4, 4, 3]
    for kt in range(1, 40):
        assert sum(partition(kt)) == kt and max(partition(kt)) <= 5


@pytest.mark.parametrize("p,k,t", [(po.P62, 3, 1), (433, 3, 1), (po.P62, 1, 1), (po.P62, 2, 2), (746497, 2, 1),
                                   (po.P62, 1, 3), (po.P62, 4, 0), (5038849, 3, 1), (po.P62, 3, 4), (433, 3, 4),
                                   (po.P62, 4, 1), (po.P62, 1, 4)])
def test_limb31_dot_is_exact_and_fits(p, k, t):
    rnd = random.Random(p % 1000 + k)
    if p == po.P62:
        w2, w3 = po.P62_OMEGA[8], po.P62_OMEGA[9]
    elif p == 433:
        w2, w3 = 354, 150
    else:
        w2, w3 = _roots(p, 8, 9)
    pss = po.PackedSecretSharing(t, 8, k, p, w2, w3)
    Mx = pss.share_matrix()
    ctx = prep(p, Mx)
    special = [0, 1, p - 1, (p - 1) // 2, (p + 1) // 2, (1 << 61) % p, ((1 << 61) - (1 << 30)) % p, ((1 << 61) + (1 << 30)) % p]
    for _ in range(1500):
        v = [rnd.choice(special + [rnd.randrange(p)] * 4) for _ in range(k + t)]
        want = [sum(a * b for a, b in zip(row, v)) % p for row in Mx]
        assert [share(ctx, ctx["M"][j], v) for j in range(8)] == want
    # adversarial matrix entries at the edge of the centred range, all terms aligned
    rinv = pow(1 << 62, -1, p)
    for mr in {(p - 1) // 2, -((p - 1) // 2), 1, -1, 0, min((1 << 61) - 1, (p - 1) // 2), -min((1 << 61) - 1, (p - 1) // 2)}:
        for width in (7, 5, 10, 15):
            row = [bal(mr)] * width
            for x in special:
                v = [x] * width
                assert share(ctx, row, v) == sum((mr * rinv) % p * y for y in v) % p


def test_largest_modulus():
    """p just below 2^62 (the library accepts any odd prime < 2^62)."""
    p = (1 << 62) - 57        # prime
    assert po.PackedSecretSharing  # noqa
    rnd = random.Random(0)
    ctx = prep(p, [[rnd.randrange(p) for _ in range(4)] for _ in range(4)])
    rinv = pow(1 << 62, -1, p)
    ctx15 = prep(p, [[rnd.randrange(p) for _ in range(15)] for _ in range(2)])
    for _ in range(300):
        v = [rnd.choice([0, p - 1, (p - 1) // 2, (p + 1) // 2, rnd.randrange(p)]) for _ in range(15)]
        for row in ctx15["M"]:
            assert share(ctx15, row, v) == sum(((m1 * B + m0) * rinv) % p * x for (m0, m1), x in zip(row, v)) % p
    for _ in range(2000):
        v = [rnd.choice([0, p - 1, (p - 1) // 2, (p + 1) // 2, rnd.randrange(p)]) for _ in range(4)]
        for row in ctx["M"]:
            want = sum(((m1 * B + m0) * rinv) % p * x for (m0, m1), x in zip(row, v)) % p
            assert share(ctx, row, v) == want
