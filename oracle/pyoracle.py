"""Python big-int ORACLE for the SDA secret-sharing / masking hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path (``sda_amd`` +
``libsda_hip.so``) never imports, links or executes anything from here.

What this file is
-----------------
A plain-Python (arbitrary precision ``int``) restatement of the reference's
``client::crypto`` sharing/masking algorithms, function by function, each citing the
reference ``file:line`` it follows (paths relative to ``/root/reference``).  It is
the *slow, obviously-right* oracle used (a) to generate the committed golden
fixtures under ``tests/golden`` and (b) to pin the fast C oracle
(``oracle/sda_oracle.c``).  Loops are pure Python, so use it for small cases.

Two value modes:

* ``rust_signed`` - reproduces Rust's truncated remainder ``%`` (sign follows the
  dividend), so intermediate values live in (-q, q) exactly as in the reference.
* ``canonical``   - mathematical Z_q, values in [0, q).  This is what the HIP path
  emits; equality with the reference is *mod q* for intermediates and bit-for-bit
  after ``RecipientOutput::positive``.

Third-party algorithms that are NOT in /root/reference
------------------------------------------------------
* ``threshold-secret-sharing = "0.2"`` (client/Cargo.toml:15; no Cargo.lock).  The
  packed scheme (``share``/``reconstruct``, radix-2/radix-3 FFT, Newton
  interpolation) is restated below from the published crate's algorithm
  ("[recalled]") and cross-checked here against a direct
  DFT / Lagrange formulation and the crate's own unit-test vectors (B1, B2).
* ``rand = "0.3"`` (client/Cargo.toml:18): ``ChaChaRng`` + ``gen_range`` restated
  from the published algorithm (Appendix C); the ChaCha20 block function is
  checked against the RFC 7539 zero-key keystream.

PARITY PINNING STATUS: the reference's own tests pin only the end-to-end linear
invariant (full_loop.rs F1-F4 => [2,4,6,8]; README walkthrough F0).  Share-level
values, the ``tss`` evaluation-point convention and the ChaCha->mask mapping are
**parity unpinned** by any test or fixture inside /root/reference (share values are
non-deterministic there: OsRng).  They are pinned here only by the recalled crate
KATs (B1/B2/C1) and by the algebraic cross-checks in ``tests/test_oracle.py``.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

MASK32 = 0xFFFFFFFF
MASK64 = 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------------------
# Rust integer semantics
# --------------------------------------------------------------------------------------
def trunc_rem(a: int, q: int) -> int:
    """Rust ``a % q`` for i64: truncated remainder, sign follows the dividend
    (SURVEY.md Appendix A.2)."""
    r = abs(a) % abs(q)
    return -r if a < 0 else r


def canon(a: int, q: int) -> int:
    """Canonical representative in [0, q)."""
    return a % q


def positive(values: Sequence[int], modulus: int) -> List[int]:
    """``RecipientOutput::positive`` - client/src/receive.rs:13-21."""
    return [v + modulus if v < 0 else v for v in values]


def _rem(mode: str):
    if mode == "rust_signed":
        return trunc_rem
    if mode == "canonical":
        return canon
    raise ValueError(mode)


# --------------------------------------------------------------------------------------
# numtheory of threshold-secret-sharing 0.2  [recalled, Appendix B]
# --------------------------------------------------------------------------------------
def tss_mod_pow(x: int, e: int, prime: int, rem=trunc_rem) -> int:
    """Square-and-multiply with ``%`` after every product (tss numtheory::mod_pow)."""
    acc = 1
    while e > 0:
        if e % 2 == 1:
            acc = rem(acc * x, prime)
        x = rem(x * x, prime)
        e >>= 1
    return acc


def tss_mod_inverse(k: int, prime: int) -> int:
    """tss numtheory::mod_inverse: extended Euclid, result made non-negative."""
    k2 = k + prime if k < 0 else k
    # gcd(prime, k2) -> (g, s, t) with s*prime + t*k2 = g ; inverse is t
    r0, r1, t0, t1 = prime, k2, 0, 1
    while r1 != 0:
        qq = r0 // r1
        r0, r1 = r1, r0 - qq * r1
        t0, t1 = t1, t0 - qq * t1
    assert r0 == 1, "not invertible"
    return t0 + prime if t0 < 0 else t0


def tss_fft2(a_coef: Sequence[int], omega: int, prime: int, rem=trunc_rem) -> List[int]:
    """Radix-2 recursive DFT, tss numtheory::fft2 [recalled]."""
    n = len(a_coef)
    if n == 1:
        return list(a_coef)
    b = tss_fft2(a_coef[0::2], tss_mod_pow(omega, 2, prime, rem), prime, rem)
    c = tss_fft2(a_coef[1::2], tss_mod_pow(omega, 2, prime, rem), prime, rem)
    half = n >> 1
    out = [0] * n
    for i in range(half):
        w = tss_mod_pow(omega, i, prime, rem)
        out[i] = rem(b[i] + w * c[i], prime)
        out[i + half] = rem(b[i] - w * c[i], prime)
    return out


def tss_fft2_inverse(a_point: Sequence[int], omega: int, prime: int, rem=trunc_rem) -> List[int]:
    """tss numtheory::fft2_inverse: forward FFT with omega^-1, then scale by len^-1."""
    omega_inv = tss_mod_inverse(omega, prime)
    len_inv = tss_mod_inverse(len(a_point), prime)
    scaled = tss_fft2(a_point, omega_inv, prime, rem)
    return [rem(x * len_inv, prime) for x in scaled]


def tss_fft3(a_coef: Sequence[int], omega: int, prime: int, rem=trunc_rem) -> List[int]:
    """Radix-3 recursive DFT, tss numtheory::fft3 [recalled]."""
    n = len(a_coef)
    if n == 1:
        return list(a_coef)
    w3 = tss_mod_pow(omega, 3, prime, rem)
    b = tss_fft3(a_coef[0::3], w3, prime, rem)
    c = tss_fft3(a_coef[1::3], w3, prime, rem)
    d = tss_fft3(a_coef[2::3], w3, prime, rem)
    third = n // 3
    out = [0] * n
    for i in range(third):
        for off in (0, third, 2 * third):
            w = tss_mod_pow(omega, i + off, prime, rem)
            out[i + off] = rem(b[i] + w * c[i] + w * w * d[i], prime)
    return out


def tss_newton_interpolation_general(points: Sequence[int], values: Sequence[int], prime: int,
                                     rem=trunc_rem) -> Tuple[List[int], List[int]]:
    """Divided differences mod prime (tss numtheory::newton_interpolation_general)."""
    n = len(points)
    # store[i] = (a, b, value): divided difference over points[a..=b]
    store = [(i, i, rem(values[i], prime)) for i in range(n)]
    for j in range(1, n):
        for i in range(n - 1, j - 1, -1):
            index_lower = store[i - 1][0]
            index_upper = store[i][1]
            point_lower = points[index_lower]
            point_upper = points[index_upper]
            point_diff = rem(point_upper - point_lower, prime)
            point_diff_inverse = tss_mod_inverse(point_diff, prime)
            coef_lower = store[i - 1][2]
            coef_upper = store[i][2]
            coef_diff = rem(coef_upper - coef_lower, prime)
            fraction = rem(coef_diff * point_diff_inverse, prime)
            store[i] = (index_lower, index_upper, fraction)
    return list(points), [s[2] for s in store]


def tss_newton_evaluate(poly: Tuple[List[int], List[int]], point: int, prime: int, rem=trunc_rem) -> int:
    """Evaluate a Newton-form polynomial (tss numtheory::newton_evaluate)."""
    points, coefs = poly
    newton_points = [1]
    for i in range(len(points) - 1):
        diff = rem(point - points[i], prime)
        newton_points.append(rem(newton_points[-1] * diff, prime))
    acc = 0
    for c, p in zip(coefs, newton_points):
        acc = rem(acc + c * p, prime)
    return acc


# --------------------------------------------------------------------------------------
# tss::packed::PackedSecretSharing  [recalled, Appendix B] -- struct literal at
# client/src/crypto/sharing/packed_shamir.rs:14-21
# --------------------------------------------------------------------------------------
class PackedSecretSharing:
    def __init__(self, threshold: int, share_count: int, secret_count: int, prime: int,
                 omega_secrets: int, omega_shares: int):
        self.threshold = threshold
        self.share_count = share_count
        self.secret_count = secret_count
        self.prime = prime
        self.omega_secrets = omega_secrets
        self.omega_shares = omega_shares

    def reconstruct_limit(self) -> int:
        return self.threshold + self.secret_count

    def is_fft_shape(self) -> bool:
        """True iff the parameters are expressible in tss 0.2 (FFT sizes are exact)."""
        m2 = self.threshold + self.secret_count + 1
        m3 = self.share_count + 1
        def is_pow(n, b):
            while n % b == 0 and n > 1:
                n //= b
            return n == 1
        return (is_pow(m2, 2) and is_pow(m3, 3)
                and pow(self.omega_secrets, m2, self.prime) == 1
                and pow(self.omega_shares, m3, self.prime) == 1)

    def recover_polynomial(self, secrets, randomness, rem=trunc_rem):
        values = [0] + list(secrets) + list(randomness)
        assert len(values) == self.reconstruct_limit() + 1
        return tss_fft2_inverse(values, self.omega_secrets, self.prime, rem)

    def evaluate_polynomial(self, coefficients, rem=trunc_rem):
        assert len(coefficients) == self.share_count + 1
        return tss_fft3(coefficients, self.omega_shares, self.prime, rem)

    def share_fft(self, secrets, randomness, mode="rust_signed"):
        """tss ``share`` with the OsRng draw replaced by injected ``randomness``."""
        rem = _rem(mode)
        assert len(secrets) == self.secret_count and len(randomness) == self.threshold
        poly = self.recover_polynomial(secrets, randomness, rem)
        poly = poly + [0] * (self.share_count - self.reconstruct_limit())
        evals = self.evaluate_polynomial(poly, rem)
        assert evals[0] % self.prime == 0
        return evals[1:]

    def reconstruct_newton(self, indices, shares, mode="rust_signed"):
        """tss ``reconstruct``: Newton interpolation through (1,0) + share points."""
        rem = _rem(mode)
        assert len(shares) == len(indices)
        assert len(shares) >= self.reconstruct_limit()
        points = [1] + [tss_mod_pow(self.omega_shares, i + 1, self.prime, rem) for i in indices]
        values = [0] + list(shares)
        poly = tss_newton_interpolation_general(points, values, self.prime, rem)
        return [tss_newton_evaluate(poly, tss_mod_pow(self.omega_secrets, e, self.prime, rem),
                                    self.prime, rem)
                for e in range(1, self.secret_count + 1)]

    # ---- matrix (Lagrange) formulation: what the HIP path implements -----------------
    def share_matrix(self) -> List[List[int]]:
        """M[j][i] (n x (k+t)): share_j = sum_i M[j][i] * v_i, v = secrets ++ randomness.
        M[j][i] = l_{i+1}(omega_shares^(j+1)) with l_m the Lagrange basis on the nodes
        omega_secrets^0..omega_secrets^(t+k); the column for node 0 (value 0) is dropped.
        Coincides with ``share_fft`` whenever ``is_fft_shape()``."""
        p = self.prime
        m = self.threshold + self.secret_count
        nodes = [pow(self.omega_secrets, e, p) for e in range(m + 1)]
        assert len(set(nodes)) == m + 1, "omega_secrets order too small"
        rows = []
        for j in range(self.share_count):
            x = pow(self.omega_shares, j + 1, p)
            row = []
            for i in range(1, m + 1):
                num, den = 1, 1
                for l in range(m + 1):
                    if l != i:
                        num = num * (x - nodes[l]) % p
                        den = den * (nodes[i] - nodes[l]) % p
                row.append(num * pow(den, -1, p) % p)
            rows.append(row)
        return rows

    def share_lagrange(self, secrets, randomness) -> List[int]:
        p = self.prime
        v = [s % p for s in secrets] + [r % p for r in randomness]
        return [sum(a * b for a, b in zip(row, v)) % p for row in self.share_matrix()]

    def reconstruct_matrix(self, indices) -> List[List[int]]:
        """R[e][c] (k x n'): secret_e = sum_c R[e][c]*share_c; nodes {1} U {omega_shares^(idx+1)}."""
        p = self.prime
        nodes = [1] + [pow(self.omega_shares, i + 1, p) for i in indices]
        assert len(set(nodes)) == len(nodes), "duplicate clerk index"
        rows = []
        for e in range(1, self.secret_count + 1):
            x = pow(self.omega_secrets, e, p)
            row = []
            for c in range(1, len(nodes)):
                num, den = 1, 1
                for l in range(len(nodes)):
                    if l != c:
                        num = num * (x - nodes[l]) % p
                        den = den * (nodes[c] - nodes[l]) % p
                row.append(num * pow(den, -1, p) % p)
            rows.append(row)
        return rows

    def reconstruct_lagrange(self, indices, shares) -> List[int]:
        p = self.prime
        assert len(shares) == len(indices) and len(shares) >= self.reconstruct_limit()
        return [sum(a * (b % p) for a, b in zip(row, shares)) % p
                for row in self.reconstruct_matrix(indices)]

    # ---- the library's CSPRNG share map (no reference counterpart: the reference draws from OsRng) ---------
    def share_systematic(self, secrets, draws) -> List[int]:
        """include/sda_hip.h "CSPRNG share map": the t draws ARE shares 0..t-1; the other shares are the values at
        omega_shares^(t+1..n) of the polynomial of degree <= t+k through (1, 0), (omega_secrets^i, secret_i) and
        (omega_shares^(j+1), draw_j).  Restated with tss's own Newton interpolation (the product uses a Lagrange
        matrix): canonical residues."""
        p, t, k = self.prime, self.threshold, self.secret_count
        assert len(secrets) == k and len(draws) == t
        points = ([1] + [pow(self.omega_secrets, i, p) for i in range(1, k + 1)] +
                  [pow(self.omega_shares, j + 1, p) for j in range(t)])
        assert len(set(points)) == len(points), "a share point collides with a node"
        values = [0] + [s % p for s in secrets] + [d % p for d in draws]
        poly = tss_newton_interpolation_general(points, values, p, canon)
        rest = [tss_newton_evaluate(poly, pow(self.omega_shares, j + 1, p), p, canon) for j in range(t, self.share_count)]
        return [d % p for d in draws] + rest

    def implied_tss_randomness(self, secrets, draws) -> List[int]:
        """the randomness tss's `share` would have needed to produce share_systematic(secrets, draws): the same
        polynomial's values at omega_secrets^(k+1..k+t).  share_lagrange(secrets, that) == share_systematic(secrets, draws)
        - the two parametrisations describe the same sharings."""
        p, t, k = self.prime, self.threshold, self.secret_count
        points = ([1] + [pow(self.omega_secrets, i, p) for i in range(1, k + 1)] +
                  [pow(self.omega_shares, j + 1, p) for j in range(t)])
        values = [0] + [s % p for s in secrets] + [d % p for d in draws]
        poly = tss_newton_interpolation_general(points, values, p, canon)
        return [tss_newton_evaluate(poly, pow(self.omega_secrets, k + 1 + j, p), p, canon) for j in range(t)]

    # ---- what the oracle exposes to the schemes below -----------------------------------
    def share(self, secrets, randomness, mode):
        if mode == "rust_signed" and self.is_fft_shape() and self.prime * self.prime < 2 ** 62:
            return self.share_fft(secrets, randomness, mode)
        return self.share_lagrange(secrets, randomness)

    def reconstruct(self, indices, shares, mode):
        if mode == "rust_signed" and self.is_fft_shape() and self.prime * self.prime < 2 ** 62:
            return self.reconstruct_newton(indices, shares, mode)
        return self.reconstruct_lagrange(indices, shares)


# --------------------------------------------------------------------------------------
# rand 0.3  ChaChaRng / gen_range  [recalled, Appendix C]
# --------------------------------------------------------------------------------------
CHACHA_CONST = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)


def _rotl32(x, n):
    return ((x << n) & MASK32) | (x >> (32 - n))


def _quarter_round(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl32(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl32(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl32(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl32(s[b] ^ s[c], 7)


def chacha_block(state: Sequence[int], rounds: int = 20) -> List[int]:
    """ChaCha core: ``rounds`` rounds (rounds/2 double rounds), output = working + input."""
    w = list(state)
    for _ in range(rounds // 2):
        _quarter_round(w, 0, 4, 8, 12); _quarter_round(w, 1, 5, 9, 13)
        _quarter_round(w, 2, 6, 10, 14); _quarter_round(w, 3, 7, 11, 15)
        _quarter_round(w, 0, 5, 10, 15); _quarter_round(w, 1, 6, 11, 12)
        _quarter_round(w, 2, 7, 8, 13); _quarter_round(w, 3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(w, state)]


class ChaChaRng:
    """rand 0.3 ``ChaChaRng::from_seed(&[u32])``: key = first <=8 seed words (rest 0),
    128-bit block counter in words 12..15 starting at 0, 20 rounds."""

    def __init__(self, seed_words: Sequence[int]):
        key = [0] * 8
        for i, w in enumerate(list(seed_words)[:8]):
            key[i] = w & MASK32
        self.state = list(CHACHA_CONST) + key + [0, 0, 0, 0]
        self.buffer: List[int] = []
        self.index = 16

    def set_counter(self, counter_low: int, counter_high: int):
        """rand 0.3 ``ChaChaRng::set_counter(counter_low: u64, counter_high: u64)`` [recalled]: words 12, 13 = low / high half of
        counter_low, words 14, 15 = low / high half of counter_high; buffered output is discarded.  The reference never
        calls it (from_seed starts at 0, chacha.rs:36) - the harness uses it to reach the 2^32 / 2^64 carries of the 128-bit
        block counter without drawing 256 GB."""
        self.state[12] = counter_low & MASK32
        self.state[13] = (counter_low >> 32) & MASK32
        self.state[14] = counter_high & MASK32
        self.state[15] = (counter_high >> 32) & MASK32
        self.index = 16

    def _update(self):
        self.buffer = chacha_block(self.state, 20)
        self.index = 0
        for i in (12, 13, 14, 15):  # 128-bit counter increment with carry
            self.state[i] = (self.state[i] + 1) & MASK32
            if self.state[i] != 0:
                break

    def next_u32(self) -> int:
        if self.index == 16:
            self._update()
        v = self.buffer[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        hi = self.next_u32()
        return (hi << 32) | self.next_u32()

    def gen_range_i64(self, low: int, high: int) -> int:
        """``Rng::gen_range(low, high)`` for i64 via ``Range<i64>`` (u64 zone rejection)."""
        rng_range = (high - low) & MASK64
        zone = MASK64 - (MASK64 % rng_range)
        while True:
            v = self.next_u64()
            if v < zone:
                return low + (v % rng_range)


# --------------------------------------------------------------------------------------
# client/src/crypto/sharing
# --------------------------------------------------------------------------------------
class AdditiveSecretSharing:
    """client/src/crypto/sharing/additive.rs:6-73."""

    def __init__(self, share_count: int, modulus: int, mode: str = "rust_signed"):
        self.share_count = share_count
        self.modulus = modulus
        self.mode = mode

    def batch_input_size(self):   # additive.rs:24-26
        return 1

    def batch_output_size(self):  # additive.rs:28-30
        return self.share_count

    def rand_per_batch(self):
        return self.share_count - 1

    def generate_for_batch(self, batch_input, randomness):
        """additive.rs:32-51; the OsRng draws (:42-44) are injected as ``randomness``."""
        rem = _rem(self.mode)
        if len(batch_input) != 1:
            raise ValueError("Batch input wrong length")          # additive.rs:33
        assert len(randomness) == self.share_count - 1
        secret = batch_input[0]
        shares = list(randomness)
        last = secret
        for x in shares:                                           # additive.rs:47 (fold)
            last = rem(last - x, self.modulus)
        return shares + [last]

    def reconstruct(self, indexed_shares):
        """additive.rs:55-73 - column sum mod q over the clerk vectors (indices ignored)."""
        rem = _rem(self.mode)
        dimension = len(indexed_shares[0][1]) if indexed_shares else 0
        result = [0] * dimension
        for _, shares in indexed_shares:
            if len(shares) != dimension:
                raise ValueError("Mismatching dimension")          # additive.rs:64
            for ix, s in enumerate(shares):
                result[ix] = rem(result[ix] + s, self.modulus)     # additive.rs:66-67
        return result


class PackedShamirGenerator:
    """client/src/crypto/sharing/packed_shamir.rs:6-45."""

    def __init__(self, threshold, share_count, secret_count, prime, omega_secrets, omega_shares,
                 mode="rust_signed"):
        self.pss = PackedSecretSharing(threshold, share_count, secret_count, prime,
                                       omega_secrets, omega_shares)
        self.mode = mode

    def batch_input_size(self):
        return self.pss.secret_count

    def batch_output_size(self):
        return self.pss.share_count

    def rand_per_batch(self):
        return self.pss.threshold

    def generate_for_batch(self, batch_input, randomness):
        if len(batch_input) != self.batch_input_size():            # packed_shamir.rs:41
            raise ValueError("Sharing failed for packed secret sharing scheme")
        return self.pss.share(batch_input, randomness, self.mode)  # packed_shamir.rs:42


def generate(gen, secrets: Sequence[int], randomness: Sequence[int]) -> List[List[int]]:
    """``impl<G: BatchShareGenerator> ShareGenerator for G`` - batched.rs:18-53.
    ``randomness`` is the flat draw sequence: rand_per_batch values for batch 0, then
    batch 1, ... (the order the reference would call its RNG in)."""
    k = gen.batch_input_size()
    n = gen.batch_output_size()
    t = gen.rand_per_batch()
    nb = (len(secrets) + k - 1) // k                              # batched.rs:23
    assert len(randomness) == nb * t
    out: List[List[int]] = [[] for _ in range(n)]                 # batched.rs:25-28
    for b in range(nb):
        if (b + 1) * k <= len(secrets):                           # batched.rs:33-36
            batch = list(secrets[b * k:(b + 1) * k])
        else:                                                     # batched.rs:37-43 (zero pad)
            batch = list(secrets[b * k:])
            batch += [0] * (k - len(batch))
        shares = gen.generate_for_batch(batch, randomness[b * t:(b + 1) * t])
        for j, s in enumerate(shares):                            # batched.rs:46-48
            out[j].append(s)
    return out


def combine(shares: Sequence[Sequence[int]], modulus: int, mode="rust_signed",
            ragged_msg="Wrong dimension") -> List[int]:
    """``Combiner::combine`` - combiner.rs:15-29 (also full.rs:37-52)."""
    rem = _rem(mode)
    dimension = len(shares[0]) if shares else 0                   # combiner.rs:17
    result = [0] * dimension
    for share in shares:
        if len(share) != dimension:
            raise ValueError(ragged_msg)                          # combiner.rs:21
        for ix, v in enumerate(share):
            result[ix] = rem(result[ix] + v, modulus)             # combiner.rs:23-24
    return result


class PackedShamirReconstructor:
    """client/src/crypto/sharing/packed_shamir.rs:47-89 + batched.rs:68-97."""

    def __init__(self, dimension, threshold, share_count, secret_count, prime, omega_secrets,
                 omega_shares, mode="rust_signed"):
        self.output_size = dimension
        self.pss = PackedSecretSharing(threshold, share_count, secret_count, prime,
                                       omega_secrets, omega_shares)
        self.mode = mode

    def reconstruct_for_batch(self, indices, batch_shares):
        if len(batch_shares) != len(indices):                     # packed_shamir.rs:74
            raise ValueError("Inputs must have same length")
        if len(batch_shares) < self.pss.reconstruct_limit():      # packed_shamir.rs:75
            raise ValueError("Not enough shares to reconstruct")
        return self.pss.reconstruct(indices, batch_shares, self.mode)

    def reconstruct(self, indexed_shares):
        """batched.rs:68-97."""
        indices = [i for i, _ in indexed_shares]
        k = self.pss.secret_count
        nb = (self.output_size + k - 1) // k                      # batched.rs:77
        secrets: List[int] = []
        for b in range(nb):
            col = [sh[b] for _, sh in indexed_shares]             # batched.rs:83-85
            secrets.extend(self.reconstruct_for_batch(indices, col))
        return secrets[:self.output_size]                         # batched.rs:94


# --------------------------------------------------------------------------------------
# client/src/crypto/masking
# --------------------------------------------------------------------------------------
class NoneMasker:
    """masking/none.rs:13-33."""

    def mask(self, secrets):
        return [], list(secrets)

    def combine(self, masks):
        assert all(len(m) == 0 for m in masks)                    # none.rs:23
        return []

    def unmask(self, mask, masked):
        assert len(mask) == 0                                     # none.rs:30
        return list(masked)


class FullMasker:
    """masking/full.rs:21-67; OsRng draws (:25) injected as ``randomness``."""

    def __init__(self, modulus, mode="rust_signed"):
        self.modulus = modulus
        self.mode = mode

    def mask(self, secrets, randomness):
        rem = _rem(self.mode)
        assert len(randomness) == len(secrets)
        masks = list(randomness)
        return masks, [rem(s + m, self.modulus) for s, m in zip(secrets, masks)]   # full.rs:30

    def combine(self, masks):
        # full.rs:37-52; ragged input is an assert_eq! panic (:43)
        try:
            return combine(masks, self.modulus, self.mode, ragged_msg="assert")
        except ValueError as e:  # pragma: no cover - exercised in tests
            raise AssertionError(str(e))

    def unmask(self, mask, masked):
        rem = _rem(self.mode)
        assert len(mask) == len(masked)                           # full.rs:58
        return [rem(ms - m, self.modulus) for ms, m in zip(masked, mask)]          # full.rs:62


class ChaChaMasker:
    """masking/chacha.rs:24-93."""

    def __init__(self, modulus, dimension, seed_bitsize, mode="rust_signed"):
        self.modulus = modulus
        self.dimension = dimension
        self.seed_bitsize = seed_bitsize
        self.mode = mode

    def seed_wordsize(self):
        return (self.seed_bitsize + 31) // 32                     # chacha.rs:31

    def expand(self, seed_words, count=None):
        rng = ChaChaRng([w & MASK32 for w in seed_words])         # chacha.rs:36 / :62-67
        n = self.dimension if count is None else count
        return [rng.gen_range_i64(0, self.modulus) for _ in range(n)]   # chacha.rs:37-39

    def mask(self, secrets, seed_words):
        """chacha.rs:24-54; the OsRng seed (:29-33) is injected as ``seed_words`` (u32 each)."""
        rem = _rem(self.mode)
        assert self.dimension == len(secrets)                     # chacha.rs:26
        assert len(seed_words) == self.seed_wordsize()
        mask = self.expand(seed_words, len(secrets))
        masked = [rem(s + m, self.modulus) for s, m in zip(secrets, mask)]          # chacha.rs:42-45
        return [int(w) for w in seed_words], masked               # chacha.rs:48-50 (u32 as i64)

    def combine(self, seeds):
        """chacha.rs:56-77."""
        rem = _rem(self.mode)
        result = [0] * self.dimension
        for seed in seeds:
            m = self.expand(seed)
            for i in range(self.dimension):
                result[i] = rem(result[i] + m[i], self.modulus)   # chacha.rs:70-71
        return result

    def unmask(self, mask, masked):
        rem = _rem(self.mode)
        assert len(mask) == len(masked)                           # chacha.rs:83
        return [rem(ms - m, self.modulus) for ms, m in zip(masked, mask)]           # chacha.rs:88


# --------------------------------------------------------------------------------------
# Scheme dispatch (sharing/mod.rs:35-96, masking/mod.rs:33-94) from plain dict "wire" enums
# --------------------------------------------------------------------------------------
def new_share_generator(scheme: dict, mode="rust_signed"):
    if scheme["kind"] == "Additive":
        return AdditiveSecretSharing(scheme["share_count"], scheme["modulus"], mode)
    return PackedShamirGenerator(scheme["privacy_threshold"], scheme["share_count"],
                                 scheme["secret_count"], scheme["prime_modulus"],
                                 scheme["omega_secrets"], scheme["omega_shares"], mode)


def new_secret_reconstructor(scheme: dict, dimension: int, mode="rust_signed"):
    if scheme["kind"] == "Additive":
        return AdditiveSecretSharing(scheme["share_count"], scheme["modulus"], mode)
    return PackedShamirReconstructor(dimension, scheme["privacy_threshold"], scheme["share_count"],
                                     scheme["secret_count"], scheme["prime_modulus"],
                                     scheme["omega_secrets"], scheme["omega_shares"], mode)


def sharing_modulus(scheme: dict) -> int:
    return scheme["modulus"] if scheme["kind"] == "Additive" else scheme["prime_modulus"]


def new_masker(scheme: dict, mode="rust_signed"):
    if scheme["kind"] == "None":
        return NoneMasker()
    if scheme["kind"] == "Full":
        return FullMasker(scheme["modulus"], mode)
    return ChaChaMasker(scheme["modulus"], scheme["dimension"], scheme["seed_bitsize"], mode)


# --------------------------------------------------------------------------------------
# Callers' data flow (participate.rs:52-76, clerk.rs:78-86, receive.rs:101-156)
# --------------------------------------------------------------------------------------
def full_aggregation(aggregation: dict, inputs: Sequence[Sequence[int]], mask_rand, share_rand,
                     clerk_subset=None, mode="rust_signed"):
    """Runs participate -> snapshot transposition -> clerk -> reveal for all participants.

    ``mask_rand[p]``: Full -> dim values; ChaCha -> seed words; None -> ignored.
    ``share_rand[p]``: flat randomness for ``generate``.
    Returns a dict with every intermediate so fixtures can pin each stage."""
    msch, ssch = aggregation["masking_scheme"], aggregation["committee_sharing_scheme"]
    dim = aggregation["vector_dimension"]
    masker = new_masker(msch, mode)
    gen = new_share_generator(ssch, mode)
    q = sharing_modulus(ssch)
    n = gen.batch_output_size()
    masks, maskeds, shares = [], [], []
    for p, secrets in enumerate(inputs):                           # participate.rs:52-76
        assert len(secrets) == dim                                 # participate.rs:44-46
        if msch["kind"] == "None":
            m, ms = masker.mask(secrets)
        else:
            m, ms = masker.mask(secrets, mask_rand[p])
        masks.append(m); maskeds.append(ms)
        shares.append(generate(gen, ms, share_rand[p]))
    # server snapshot: participant-major -> clerk-major (server/src/stores.rs:86-101)
    jobs = [[shares[p][c] for p in range(len(inputs))] for c in range(n)]
    clerk_sums = [combine(job, q, mode) for job in jobs]           # clerk.rs:85-86
    mask = masker.combine(masks) if msch["kind"] != "None" else []  # receive.rs:102-118
    subset = list(range(n)) if clerk_subset is None else list(clerk_subset)
    rec = new_secret_reconstructor(ssch, dim, mode)
    masked_output = rec.reconstruct([(c, clerk_sums[c]) for c in subset])   # receive.rs:140-144
    output = masker.unmask(mask, masked_output)                    # receive.rs:149-152
    return {
        "masks": masks, "masked": maskeds, "shares": shares, "clerk_sums": clerk_sums,
        "combined_mask": mask, "masked_output": masked_output, "output": output,
        "positive": positive(output, aggregation["modulus"]),       # receive.rs:13-21
    }


# --------------------------------------------------------------------------------------
# Parameters of SURVEY.md Appendix D (62-bit configs) and the reference's test shape
# --------------------------------------------------------------------------------------
P62 = 4611686006577364993
P62_OMEGA = {8: 631229665360524489, 9: 3451275676410824977,
             16: 2589100645267092065, 27: 365137883145458390}
PSS_433 = dict(kind="PackedShamir", secret_count=3, share_count=8, privacy_threshold=4,
               prime_modulus=433, omega_secrets=354, omega_shares=150)   # full_loop.rs:57-64


# --------------------------------------------------------------------------------------
# sda-drbg-v1: the PRODUCT's own on-device CSPRNG stream layout.  No reference
# counterpart (the reference draws from OsRng); restated here so that tests can reproduce
# device-generated randomness and pin the C oracle's copy of the same spec.
# --------------------------------------------------------------------------------------
def _drbg_state(key_words, I, stream, attempt):
    return (list(CHACHA_CONST) + list(key_words)
            + [I & MASK32, (I >> 32) & MASK32, stream & MASK32,
               ((stream >> 32) & 0xFFFFFF) | (attempt << 24)])


def _lemire(x, m):
    pr = x * m
    return (pr & MASK64) >= ((1 << 64) % m), pr >> 64


DRBG_PAIRED_MAX = 0x7F7F7F       # moduli up to here (m * m < 2^46): one 64-bit candidate word yields TWO draws


def _lemire_pair(x, m):
    """the PAIRED rule of sda-drbg-v1 for m <= DRBG_PAIRED_MAX: x m m = (ra m + rb) 2^64 + lo - Lemire's method with range m^2
    (accept iff lo >= 2^64 mod m^2, value hi uniform in [0, m^2)) read off in two steps, x m = ra 2^64 + l1 and l1 m = rb 2^64 + lo
    ("batched" bounded integers): (ra, rb) is a uniform pair in [0, m)^2"""
    p1 = x * m
    ra, l1 = p1 >> 64, p1 & MASK64
    p2 = l1 * m
    rb, lo = p2 >> 64, p2 & MASK64
    return lo >= ((1 << 64) % (m * m)), (ra, rb)


def _drbg_value_paired(kw, stream, b, T, i, m, rounds):
    """draws 2j and 2j + 1 of a batch come from ONE candidate word: block counter (b >> 3) * ceil(T / 2) + j, the same lane / word
    mapping as the unpaired rule; retry blocks are counted (b * ceil(T / 2) + j)"""
    T2, j = (T + 1) // 2, i >> 1
    o = chacha_block(_drbg_state(kw, (b >> 3) * T2 + j, stream, 0), rounds)
    c, e = (b & 7) >> 1, b & 1
    ok, pair = _lemire_pair((o[8 * e + c] << 32) | o[8 * e + 4 + c], m)
    a = 1
    while not ok and a < 256:
        o2 = chacha_block(_drbg_state(kw, b * T2 + j, stream, a), rounds)
        for q in range(8):
            ok, pair = _lemire_pair((o2[2 * q] << 32) | o2[2 * q + 1], m)
            if ok:
                break
        a += 1
    return pair[i & 1]


def drbg_value(key: bytes, stream: int, b: int, T: int, i: int, m: int, rounds: int = 20) -> int:
    """Draw i (of T) for batch b of stream `stream`, uniform in [0, m)."""
    kw = [int.from_bytes(key[4 * j:4 * j + 4], "little") for j in range(8)]
    if m <= DRBG_PAIRED_MAX:
        return _drbg_value_paired(kw, stream, b, T, i, m, rounds)
    o = chacha_block(_drbg_state(kw, (b >> 3) * T + i, stream, 0), rounds)
    c, e = (b & 7) >> 1, b & 1
    ok, val = _lemire((o[8 * e + c] << 32) | o[8 * e + 4 + c], m)
    a = 1
    while not ok and a < 256:
        o2 = chacha_block(_drbg_state(kw, b * T + i, stream, a), rounds)
        for j in range(8):
            ok, val = _lemire((o2[2 * j] << 32) | o2[2 * j + 1], m)
            if ok:
                break
        a += 1
    return val


def drbg_call_key(master: bytes, call_index: int) -> bytes:
    """sda-drbg-v1 call key: words 0..7 of ChaCha20(key = master, counter = call index, nonce = "sdak" "dfv1")."""
    kw = [int.from_bytes(master[4 * j:4 * j + 4], "little") for j in range(8)]
    st = list(CHACHA_CONST) + kw + [call_index & MASK32, (call_index >> 32) & MASK32, 0x6b616473, 0x31766664]
    o = chacha_block(st, 20)
    return b"".join(int(w).to_bytes(4, "little") for w in o[:8])


def drbg_fill(key: bytes, stream: int, batches: int, T: int, m: int, rounds: int = 20) -> List[int]:
    return [drbg_value(key, stream, b, T, i, m, rounds) for b in range(batches) for i in range(T)]


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def synthetic_secret(seed: int, participant: int, i: int, modulus: int) -> int:
    """SURVEY.md 8d synthetic input."""
    return splitmix64(seed ^ (((participant << 32) | i) & MASK64)) % modulus


# --------------------------------------------------------------------------------------
# Share-vector wire codec: integer-encoding 1.0 `VarInt for i64`
# [recalled] as used by client/src/crypto/encryption/sodium.rs:36-41 (encode) and :83-89
# (decode): zig-zag, then LEB128 (7-bit groups, least significant first, MSB = continuation).
# --------------------------------------------------------------------------------------
def varint_encode_i64(v: int) -> bytes:
    """`share.encode_var(&mut buf)` - sodium.rs:39"""
    assert -(1 << 63) <= v < (1 << 63)
    n = ((v << 1) ^ (v >> 63)) & MASK64                 # zig-zag
    out = bytearray()
    while n >= 0x80:
        out.append(0x80 | (n & 0x7F))
        n >>= 7
    out.append(n)
    return bytes(out)


def varint_encode(values: Sequence[int]) -> bytes:
    """sodium.rs:36-41: the encodings concatenated."""
    return b"".join(varint_encode_i64(v) for v in values)


def varint_decode_one(src: bytes) -> Tuple[int, int]:
    """`Share::decode_var(reader)` - sodium.rs:86; u64::decode_var + zig-zag decode."""
    result, shift = 0, 0
    for b in src:
        result |= ((b & 0x7F) << shift) & MASK64
        shift += 7
        if b & 0x80 == 0 or shift > 10 * 7:
            break
    v = (result >> 1) ^ (-(result & 1) & MASK64)
    if v >= 1 << 63:
        v -= 1 << 64
    return v, shift // 7


def varint_decode(raw: bytes) -> List[int]:
    """sodium.rs:83-89: decode until the reader is empty."""
    out, pos = [], 0
    while pos < len(raw):
        v, size = varint_decode_one(raw[pos:])
        out.append(v)
        pos += size
    return out
