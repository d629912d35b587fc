"""Big-int model of the transform form of packed-Shamir share generation (sda_amd/csrc/fft_kernels.hip, round 3): tss's
own algorithm - radix-2 inverse transform over the k+t+1 secret nodes, zero-extension, radix-3 forward transform over the
n+1 share points (packed_shamir.rs:42 -> tss `share`) - on lazily reduced UNSIGNED 64-bit values with
Shoup multiplications by table constants, in exactly the kernel's order: a single radix-2 level when their number is odd,
then radix-4 passes; the first two radix-3 levels folded into the zero-extending scatter; a single radix-3 level when the
rest is odd, then radix-9 passes.  Checks exactness against the oracle's FFT / matrix / Lagrange forms and that every
intermediate fits the 64-bit register (and the [0, 2p) / [0, 4p) range) the kernel keeps it in.  CPU only."""
import random

import pytest

from oracle import pyoracle as po

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def u64(x):
    assert 0 <= x <= M64, x                                      # the kernel's registers: no wrap-around anywhere
    return x


class Dev:
    """device-side arithmetic with register-width and range assertions"""

    def __init__(self, p):
        assert 2 <= p < (1 << 62)                                # every modulus the library takes (make_mod): 4p < 2^64
        self.p, self.p2, self.np = p, 2 * p, (1 << 64) - p

    def pair(self, w):
        """host side: a constant and its companion floor(w 2^64 / p)"""
        w %= self.p
        return w, (w << 64) // self.p

    def mulS(self, x, c):
        """x (ANY 64-bit value) times the constant c = (w, ws): congruent to x w, in [0, 2p)"""
        w, ws = c
        u64(x)
        q = (x * ws) >> 64
        r = x * w - q * self.p
        assert 0 <= r < self.p2, (x, w, r)
        # the kernel computes the low 64 bits as x w + q (2^64 - p): one pair of accumulating 32 x 32 products plus four
        # low products into the high word
        x0, x1, q0, q1 = x & M32, x >> 32, q & M32, q >> 32
        w0, w1, n0, n1 = w & M32, w >> 32, self.np & M32, self.np >> 32
        t = (x0 * w0 + q0 * n0) & M64
        h = (x0 * w1 + x1 * w0 + q0 * n1 + q1 * n0) & M32
        assert (t + (h << 32)) & M64 == r
        return r

    def csub(self, x, m):
        """x < 2m -> x < m (the borrow of the 64-bit subtraction selects)"""
        u64(x)
        assert x < 2 * m, (x, m)
        return x - m if x >= m else x

    def red2(self, x):
        return self.csub(x, self.p2)

    def r3(self, A, Bv, Cv, om):
        """radix-3 butterfly on A, B, C in [0, 2p): y_d = A + w^d B + w^2d C, in [0, 4p)"""
        p2 = self.p2
        assert A < p2 and Bv < p2 and Cv < p2
        w = self.mulS(u64(Bv + p2 - Cv), om)
        y0 = u64(self.red2(u64(A + Bv)) + Cv)
        y1 = u64(self.red2(u64(A + p2 - Cv)) + w)
        y2 = u64(self.red2(u64(A + p2 - Bv)) + (p2 - w))
        for y in (y0, y1, y2):
            assert y < 2 * p2
        return y0, y1, y2


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
    p, p2 = dev.p, dev.p2
    m2, m3 = k + t + 1, n + 1
    a, b = m2.bit_length() - 1, 0
    while 3 ** b < m3:
        b += 1
    assert 1 << a == m2 and 3 ** b == m3 and b >= 2
    w2i = pow(w2, -1, p)
    tw2 = [dev.pair(pow(w2i, j, p)) for j in range(max(m2 // 2, 1))]
    tw3 = [dev.pair(pow(w3, j, p)) for j in range(m3)]
    om = dev.pair(pow(w3, m3 // 3, p))
    scale = dev.pair(pow(m2, -1, p))
    x = [0] + [s % p for s in secrets] + [r % p for r in draws]            # canonical
    # ---- radix-2 inverse transform, decimation in frequency: natural order in, bit-reversed order out, values in [0, 2p)
    mblk, lg = m2, a
    if lg & 1:
        h = mblk // 2
        for jj in range(h):
            av, bv = x[jj], x[jj + h]
            x[jj] = dev.red2(u64(av + bv))
            x[jj + h] = dev.mulS(u64(av + p2 - bv), tw2[jj])
        mblk //= 2
        lg -= 1
    while lg >= 2:
        qd, step = mblk // 4, m2 // mblk
        for i in range(m2 // 4):
            blk, jj = i // qd, i % qd
            base = blk * mblk + jj
            x0, x1, x2, x3 = (x[base + e * qd] for e in range(4))
            assert max(x0, x1, x2, x3) < p2
            a0, a1 = dev.red2(u64(x0 + x2)), dev.red2(u64(x1 + x3))
            a3 = dev.mulS(u64(x1 + p2 - x3), tw2[(jj + qd) * step])
            if qd > 1:
                a2 = dev.mulS(u64(x0 + p2 - x2), tw2[jj * step])
                b1 = dev.mulS(u64(a0 + p2 - a1), tw2[2 * jj * step])
                b3 = dev.mulS(u64(a2 + p2 - a3), tw2[2 * jj * step])
            else:
                a2 = dev.red2(u64(x0 + p2 - x2))
                b1 = dev.red2(u64(a0 + p2 - a1))
                b3 = dev.red2(u64(a2 + p2 - a3))
            x[base], x[base + qd], x[base + 2 * qd], x[base + 3 * qd] = dev.red2(u64(a0 + a1)), b1, dev.red2(u64(a2 + a3)), b3
        lg -= 2
        mblk //= 4
    assert lg == 0
    # ---- scale by 1 / m2, zero-extend, first two radix-3 levels (decimation in time, digit-reversed input) ----------------
    y = [None] * m3
    ninth, S1 = m3 // 9, m3 // 3
    S2 = ninth
    nz = [[e1 * S2 + e0 * S1 < m2 for e1 in range(3)] for e0 in range(3)]   # the plan's nz_mask, [e0][e1]
    for q in range(ninth):
        r = trirev(q, b - 2)
        v = [[None] * 3 for _ in range(3)]
        for e1 in range(3):
            inp = [0, 0, 0]
            for e0 in range(3):
                ci = r + e1 * S2 + e0 * S1
                if nz[e0][e1] and ci < m2:
                    inp[e0] = dev.mulS(x[bitrev(ci, a)], scale)
                else:
                    assert ci >= m2                                        # the mask never hides a coefficient
            if nz[1][e1] or nz[2][e1]:
                v[e1] = [dev.red2(z) for z in dev.r3(inp[0], inp[1], inp[2], om)]
            else:
                v[e1] = [inp[0]] * 3
        for jj in range(3):
            Bv, Cv = v[1][jj], v[2][jj]
            if jj:
                Bv = dev.mulS(Bv, tw3[jj * ninth])
                Cv = dev.mulS(Cv, tw3[2 * jj * ninth])
            y[9 * q + jj], y[9 * q + jj + 3], y[9 * q + jj + 6] = dev.r3(v[0][jj], Bv, Cv, om)
    assert all(z is not None and z < 2 * p2 for z in y)
    # ---- remaining levels: a single one when their number is odd, then two at a time -----------------------------------
    t3, left = 9, b - 2
    if left & 1:
        step = m3 // (3 * t3)
        for q in range(S1):
            blk, jj = q // t3, q % t3
            base = blk * 3 * t3 + jj
            A = dev.red2(y[base])
            Bv, Cv = dev.mulS(y[base + t3], tw3[jj * step]), dev.mulS(y[base + 2 * t3], tw3[2 * jj * step])
            y[base], y[base + t3], y[base + 2 * t3] = dev.r3(A, Bv, Cv, om)
        t3 *= 3
        left -= 1
    while left:
        step_a, step_b = m3 // (3 * t3), m3 // (9 * t3)
        for q in range(ninth):
            blk, jj = q // t3, q % t3
            base = blk * 9 * t3 + jj
            av = [y[base + e * t3] for e in range(9)]
            v = [None] * 9
            for e1 in range(3):
                A = dev.red2(av[3 * e1])
                Bv, Cv = dev.mulS(av[3 * e1 + 1], tw3[jj * step_a]), dev.mulS(av[3 * e1 + 2], tw3[2 * jj * step_a])
                v[3 * e1:3 * e1 + 3] = dev.r3(A, Bv, Cv, om)
            for d in range(3):
                jb = jj + d * t3
                A = dev.red2(v[d])
                Bv, Cv = dev.mulS(v[3 + d], tw3[jb * step_b]), dev.mulS(v[6 + d], tw3[2 * jb * step_b])
                y[base + d * t3], y[base + (d + 3) * t3], y[base + (d + 6) * t3] = dev.r3(A, Bv, Cv, om)
        t3 *= 9
        left -= 2
    assert t3 == m3
    out = [dev.csub(dev.red2(z), p) for z in y]                                    # canonical
    assert out[0] == 0                                                             # f(1) = 0 (tss asserts the same)
    return out[1:]


def _roots(p, o2, o3):
    g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    return pow(g, (p - 1) // o2, p), pow(g, (p - 1) // o3, p)


@pytest.mark.parametrize("p,k,t,n", [(433, 3, 4, 8), (po.P62, 3, 4, 8), (po.P62, 8, 7, 26), (po.P62, 1, 2, 8),
                                     (po.P62, 20, 11, 80), (746497, 100, 155, 728), (po.P62, 40, 23, 242),
                                     (po.P62, 70, 57, 242), (po.P62, 2, 1, 26), (po.P62, 1, 0, 8)])
def test_transform_share_equals_the_oracle(p, k, t, n):
    """shapes cover: an odd and an even number of radix-2 levels (single level + radix-4 passes), 0 / 1 / 2 / 3 / 4 radix-3
    levels after the folded two (single level, radix-9 passes), and every zero-extension pattern of the first levels
    (m2 <= m3/9, m3/9 < m2 <= m3/3, m3/3 < m2 <= 2 m3/3 incl. tss's 256 of 729)"""
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
            # large prime, large shape: Lagrange evaluation at three share points (independent of any transform)
            lag = pss.share_lagrange(s, r)
            for j in (1, n // 2, n):
                assert got[j - 1] == lag[j - 1]


def _prime_below(x):
    x -= 1 - (x & 1)
    while not all(pow(a, x - 1, x) == 1 for a in (2, 3, 5, 7, 11, 13, 17)):
        x -= 2
    return x


@pytest.mark.parametrize("p", [_prime_below(1 << 62), po.P62, 433, 2])
def test_multiplication_and_butterfly_ranges_at_the_edges(p):
    """the Shoup product accepts ANY 64-bit operand and lands in [0, 2p); the butterfly keeps every sum inside 64 bits -
    for the largest modulus the library admits (p just below 2^62: 4p just below 2^64) and for tiny ones"""
    dev = Dev(p)
    rnd = random.Random(p & 0xFFFF)
    consts = [0, 1, p - 1, p // 2] + [rnd.randrange(p) for _ in range(40)]
    xs = [0, 1, p - 1, p, 2 * p - 1, 2 * p, 4 * p - 1, M64, M64 - 1, 1 << 63, (1 << 63) - 1] + [rnd.getrandbits(64) for _ in range(60)]
    for w in consts:
        c = dev.pair(w)
        for x in xs:
            r = dev.mulS(x & M64, c)
            assert r % p == (x & M64) * w % p
    om = dev.pair(consts[5])                                   # range checks do not need a true cube root
    edge = [0, 1, 2 * p - 1, p, p - 1] + [rnd.randrange(2 * p) for _ in range(8)]
    for A in edge:
        for Bv in edge:
            for Cv in edge:
                y = dev.r3(A, Bv, Cv, om)
                w = om[0]
                assert y[0] % p == (A + Bv + Cv) % p
                assert y[1] % p == (A - Cv + w * (Bv - Cv)) % p
                assert y[2] % p == (A - Bv - w * (Bv - Cv)) % p
