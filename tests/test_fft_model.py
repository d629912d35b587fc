"""Big-int model of the transform form of packed-Shamir share generation (sda_amd/csrc/fft_kernels.hip): tss's own
algorithm - radix-2 inverse transform over the k+t+1 secret nodes, zero-extension, radix-3 forward transform over the
n+1 share points (packed_shamir.rs:42 -> tss `share`) - on lazily reduced signed 64-bit values with
single balanced-31-bit-limb Montgomery multiplications.  Checks exactness against the oracle's FFT and matrix forms
and that every intermediate fits the registers the kernel keeps it in.  CPU only."""
import random

import pytest

from oracle import pyoracle as po
from test_limb31_model import B, MB, bal, i32, i64, sext31


def packc(c, p):
    """host side: a twiddle in Montgomery form (R = 2^62), centred, as balanced limbs"""
    mr = (c << 62) % p
    if mr > (p - 1) // 2:
        mr -= p
    return bal(mr)


class Dev:
    """device-side arithmetic with register-width assertions"""

    def __init__(self, p):
        assert p % 2 == 1 and p < (1 << 62) - (1 << 31)          # the transform path's precondition (host-checked)
        self.p, self.pinvB, self.p0, self.p1 = p, (-pow(p, -1, B)) % B, p % B, p >> 31

    def split(self, x):
        assert abs(x) <= self.p                                  # every multiplication input is a narrowed value
        x0, x1 = bal(x)
        i32(x0); i32(x1)
        return x0, x1

    def mulc(self, x, c):
        """x in [-p, p] times the constant c = (m0, m1): result in (-p, p), congruent to x * const"""
        p = self.p
        m0, m1 = c
        x0, x1 = self.split(i64(x))
        C0 = i64(m0 * x0); C1 = i64(i64(m0 * x1) + m1 * x0); C2 = i64(m1 * x1)
        q0 = sext31(((C0 & 0xFFFFFFFF) * self.pinvB) & 0xFFFFFFFF)
        C0 = i64(C0 + q0 * self.p0)
        assert C0 % B == 0
        E = i64((C0 >> 31) + q0 * self.p1)
        q1 = sext31((((C1 & 0xFFFFFFFF) + (E & 0xFFFFFFFF)) * self.pinvB) & 0xFFFFFFFF)
        E = i64(E + q1 * self.p0)
        assert (C1 + E) % B == 0
        r = i64(i64(C2 + q1 * self.p1) + (C1 >> 31) + (i64(E + MB) >> 31))
        assert -p < r < p, (r, p)
        return r

    def narrow2(self, x):
        """[-2p, 2p) -> [-p, p)"""
        i64(x)
        assert -2 * self.p <= x < 2 * self.p
        return x - self.p if x >= 0 else x + self.p


def bitrev(i, bits):
    return int(format(i, f"0{bits}b")[::-1], 2) if bits else 0


def trirev(i, digits):
    r = 0
    for _ in range(digits):
        r = r * 3 + i % 3
        i //= 3
    return r


def share_transform(dev, k, t, n, w2, w3, secrets, draws):
    """one batch, exactly as the kernel does it"""
    p = dev.p
    m2, m3 = k + t + 1, n + 1
    a, b = m2.bit_length() - 1, 0
    while 3 ** b < m3:
        b += 1
    assert 1 << a == m2 and 3 ** b == m3
    w2i = pow(w2, -1, p)
    tw2 = [packc(pow(w2i, j, p), p) for j in range(max(m2 // 2, 1))]
    tw3 = [packc(pow(w3, j, p), p) for j in range(m3)]
    omega = packc(pow(w3, m3 // 3, p), p)
    scale = packc(pow(m2, -1, p), p)
    centre = lambda v: v - p if v > (p - 1) // 2 else v
    x = [0] + [centre(s % p) for s in secrets] + [centre(r % p) for r in draws]
    # radix-2 inverse transform, decimation in frequency: natural order in, bit-reversed order out
    m = m2
    while m >= 2:
        h = m // 2
        for k0 in range(0, m2, m):
            for j in range(h):
                u, v = x[k0 + j], x[k0 + j + h]
                x[k0 + j] = dev.narrow2(u + v)
                d = dev.narrow2(u - v)
                x[k0 + j + h] = dev.mulc(d, tw2[j * (m2 // m)]) if j else d           # w^0 = 1: no multiply
        m = h
    # zero-extension: coefficient j (at bit-reversed position), scaled by 1 / m2, to its digit-reversed position
    y = [0] * m3
    for j in range(m2):
        y[trirev(j, b)] = dev.mulc(x[bitrev(j, a)], scale)
    # radix-3 forward transform, decimation in time: digit-reversed order in, natural order out
    m = 3
    while m <= m3:
        t3 = m // 3
        step = m3 // m
        for k0 in range(0, m3, m):
            for j in range(t3):
                A, Bv, C = y[k0 + j], y[k0 + j + t3], y[k0 + j + 2 * t3]
                if j:
                    Bv = dev.mulc(Bv, tw3[j * step])
                    C = dev.mulc(C, tw3[2 * j * step])
                u = dev.mulc(dev.narrow2(Bv - C), omega)
                # three-term sums in two steps: 3p does not fit a signed 64-bit register when p is close to 2^62
                y[k0 + j] = dev.narrow2(dev.narrow2(A + Bv) + C)
                y[k0 + j + t3] = dev.narrow2(dev.narrow2(A - C) + u)
                y[k0 + j + 2 * t3] = dev.narrow2(dev.narrow2(A - Bv) - u)
        m *= 3
    assert y[0] % p == 0                                                           # f(1) = 0 (tss asserts the same)
    return [v % p for v in y[1:]]


def _roots(p, o2, o3):
    g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    return pow(g, (p - 1) // o2, p), pow(g, (p - 1) // o3, p)


@pytest.mark.parametrize("p,k,t,n", [(433, 3, 4, 8), (po.P62, 3, 4, 8), (po.P62, 8, 7, 26), (po.P62, 1, 2, 8),
                                     (po.P62, 20, 11, 80), (746497, 100, 155, 728), (po.P62, 40, 23, 242)])
def test_transform_share_equals_the_oracle(p, k, t, n):
    rnd = random.Random(k * 1000 + n)
    if p == 433:
        w2, w3 = 354, 150
    elif p == 746497 and n == 728:
        w2, w3 = 95660, 610121                                                      # tss PSS_155_728_100 [recalled]
    else:
        w2, w3 = _roots(p, k + t + 1, n + 1)
    dev = Dev(p)
    pss = po.PackedSecretSharing(t, n, k, p, w2, w3)
    assert pss.is_fft_shape()
    special = [0, 1, p - 1, (p - 1) // 2, (p + 1) // 2]
    reps = (1 if n > 300 else 3) if n > 100 else 40
    Mx = pss.share_matrix() if n <= 100 else None
    for it in range(reps):
        if it == 0:
            s, r = [p - 1] * k, [p - 1] * t
        elif it == 1:
            s, r = [(p - 1) // 2] * k, [(p + 1) // 2] * t
        else:
            s = [rnd.choice(special + [rnd.randrange(p)] * 3) for _ in range(k)]
            r = [rnd.choice(special + [rnd.randrange(p)] * 3) for _ in range(t)]
        got = share_transform(dev, k, t, n, w2, w3, s, r)
        want = [v % p for v in pss.share_fft(s, r, "canonical")] if p < (1 << 31) else None
        if Mx is not None:
            mat = [sum(a * b for a, b in zip(row, s + r)) % p for row in Mx]
            assert got == mat
            if want is not None:
                assert want == mat
        elif want is not None:
            assert got == want
        else:
            # large prime, large shape: check through the polynomial itself - interpolate nothing, evaluate the
            # transform's own coefficients would be circular; use Lagrange evaluation at three share points instead
            for j in (1, n // 2, n):
                assert got[j - 1] == pss.share_lagrange(s, r)[j - 1]


def test_mulc_range_edges():
    """the single multiplication accepts any |x| <= p and lands in (-p, p), for the largest admissible modulus
    (p < 2^62 - 2^31 keeps the high limb of x inside a signed 32-bit register)"""
    p = (1 << 62) - (1 << 31) - 69
    while not all(pow(a, p - 1, p) == 1 for a in (2, 3, 5, 7, 11, 13)):
        p -= 2
    dev = Dev(p)
    rnd = random.Random(7)
    rinv = pow(1 << 62, -1, p)
    for mr in [(p - 1) // 2, -((p - 1) // 2), 1, -1, 0] + [rnd.randrange(-(p // 2), p // 2) for _ in range(50)]:
        c = bal(mr)
        for x in [p, -p, p - 1, 1 - p, (1 << 61), -(1 << 61), 0, 1, -1] + [rnd.randrange(-p, p + 1) for _ in range(50)]:
            r = dev.mulc(x, c)
            assert (r - x * mr * rinv) % p == 0
