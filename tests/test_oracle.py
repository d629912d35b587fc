"""CPU tests of the ORACLE itself: pin it against every golden vector / KAT available for this path
 and cross-check its two implementations (Python big-int vs C) and its two
formulations (tss FFT/Newton vs Lagrange matrix)."""
import itertools
import json
import os
import random

import numpy as np
import pytest

from conftest import load_golden
from oracle import coracle, pyoracle as po

P62 = po.P62


# ---- golden vectors held by the reference's own tests ----------------------------------------------------
def test_reference_end_to_end_vectors():
    """F0-F4 (full_loop.rs:29-67 -> [2,4,6,8]; README.md:157 -> 0 2 2 4 4 6 6 8 8 10): recompute every
    committed scenario from its recorded randomness in both value modes."""
    for sc in load_golden("full_loop.json")["scenarios"]:
        for mode in ("rust_signed", "canonical"):
            r = po.full_aggregation(sc["aggregation"], sc["inputs"], sc["mask_rand"], sc["share_rand"],
                                    sc["clerk_subset"], mode)
            assert r["positive"] == sc["expected_positive"], sc["name"]
            assert r == sc["stages"][mode], sc["name"]
    names = [s["name"] for s in load_golden("full_loop.json")["scenarios"]]
    assert names[:5] == ["F0_readme_walkthrough", "F1_simple", "F2_with_fullmask", "F3_with_chachamask",
                         "F4_with_packedshamir"]


def test_full_loop_any_randomness():
    """The reference uses OsRng: the revealed output must not depend on the draws (F1-F4)."""
    rnd = random.Random(7)
    add = dict(kind="Additive", share_count=3, modulus=433)
    for msk, shr in itertools.product(
            [dict(kind="None"), dict(kind="Full", modulus=433), dict(kind="ChaCha", modulus=433, dimension=4, seed_bitsize=128)],
            [add, dict(po.PSS_433)]):
        a = dict(vector_dimension=4, modulus=433, masking_scheme=msk, committee_sharing_scheme=shr)
        gen = po.new_share_generator(shr)
        nb = (4 + gen.batch_input_size() - 1) // gen.batch_input_size()
        for _ in range(5):
            mr = [[rnd.randrange(433) for _ in range(4)] if msk["kind"] == "Full" else [rnd.getrandbits(32) for _ in range(4)]
                  for _ in range(2)]
            sr = [[rnd.randrange(432) for _ in range(nb * gen.rand_per_batch())] for _ in range(2)]
            for mode in ("rust_signed", "canonical"):
                assert po.full_aggregation(a, [[1, 2, 3, 4]] * 2, mr, sr, None, mode)["positive"] == [2, 4, 6, 8]


# ---- third-party KATs (tss 0.2, rand 0.3 / RFC 7539) -----------------------------------------------------------
def test_tss_kats():
    k = load_golden("kats.json")["kats"]
    pss = po.PackedSecretSharing(4, 8, 3, 433, 354, 150)
    b1 = k["B1_recover_polynomial"]
    coeffs = pss.recover_polynomial([1, 2, 3], [8, 8, 8, 8])
    assert coeffs == b1["recalled_signed"] == b1["signed"]          # tss test_recover_polynomial, exact signs
    assert [c % 433 for c in coeffs] == b1["recalled_canonical"]
    pss26 = po.PackedSecretSharing(4, 26, 3, 433, 354, 17)
    ev = pss26.evaluate_polynomial(coeffs + [0] * 19)
    assert [e % 433 for e in ev] == k["B2_evaluate_polynomial"]["recalled_canonical"]   # tss test_evaluate_polynomial
    # independent re-derivation: direct DFT
    assert [e % 433 for e in ev] == [sum(c * pow(17, i * j, 433) for j, c in enumerate(coeffs)) % 433 for i in range(27)]
    assert [s % 433 for s in pss.share_fft([1, 2, 3], [8, 8, 8, 8])] == k["B3_share"]["expected"]
    assert pss.share_lagrange([1, 2, 3], [8, 8, 8, 8]) == k["B3_share"]["expected"]
    assert pss.share_matrix()[0] == k["B4_share_matrix_row0"]["expected"]
    for idx in k["B5_reconstruct"]["index_sets"]:
        sh = [k["B5_reconstruct"]["shares"][i] for i in idx]
        assert [v % 433 for v in pss.reconstruct_newton(idx, sh)] == [1, 2, 3]
        assert pss.reconstruct_lagrange(idx, sh) == [1, 2, 3]


def test_chacha_kats():
    k = load_golden("kats.json")["kats"]
    blk = po.chacha_block(list(po.CHACHA_CONST) + [0] * 12)
    assert [hex(w) for w in blk[:4]] == k["C1_chacha20_zero_key_block0"]["expected_first4"]   # RFC 7539 keystream
    # RFC 7539 section 2.3.2 block function test vector (key 00..1f, counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00)
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    st = list(po.CHACHA_CONST) + key + [1, 0x09000000, 0x4a000000, 0]
    out = po.chacha_block(st)
    assert out[:4] == [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3]
    assert out[12:] == [0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert list(coracle.chacha_block(st)) == out
    assert po.ChaChaMasker(433, 8, 128).expand([0, 0, 0, 0]) == k["C2_masks_seed0"]["expected"]
    assert po.ChaChaMasker(433, 8, 128).expand([1, 2, 3, 4]) == k["C3_masks_seed1234"]["expected"]
    assert hex(po.ChaChaRng([1, 2, 3, 4]).next_u64()) == k["C3_masks_seed1234"]["expected_first_u64"]
    assert list(coracle.chacha_expand([1, 2, 3, 4], 433, 8)) == k["C3_masks_seed1234"]["expected"]


def test_chacha_counter_carry_and_rejection():
    rng = po.ChaChaRng([5])
    rng.state[12] = 0xFFFFFFFF
    rng._update()
    assert rng.state[12] == 0 and rng.state[13] == 1          # 128-bit counter carries
    q = (1 << 61) + 1                                         # ~12% of candidates are rejected
    n = 2000
    want = po.ChaChaMasker(q, n, 128).expand([7, 7, 7, 7])
    r2 = po.ChaChaRng([7, 7, 7, 7])
    zone = po.MASK64 - (po.MASK64 % q)
    vals, rejected = [], 0
    while len(vals) < n:
        v = r2.next_u64()
        if v < zone:
            vals.append(v % q)
        else:
            rejected += 1
    assert rejected > 100 and vals == want
    assert list(coracle.chacha_expand([7, 7, 7, 7], q, n)) == want


# ---- FFT / Newton (tss) == Lagrange matrix (what the HIP path implements) -------------------------------------
@pytest.mark.parametrize("prime,w2,w3,t,k,n", [(433, 354, 150, 4, 3, 8), (433, 354, 17, 4, 3, 26),
                                               (746497, 95660, 610121, 155, 100, 728)])
def test_tss_fft_equals_matrix_form(prime, w2, w3, t, k, n):
    pss = po.PackedSecretSharing(t, n, k, prime, w2, w3)
    assert pss.is_fft_shape()
    rnd = random.Random(prime)
    reps = 1 if n > 100 else 20
    M = pss.share_matrix()
    for _ in range(reps):
        s = [rnd.randrange(prime) for _ in range(k)]
        r = [rnd.randrange(prime - 1) for _ in range(t)]
        fft = pss.share_fft(s, r, "rust_signed")
        assert [v % prime for v in fft] == [sum(a * b for a, b in zip(row, s + r)) % prime for row in M]
        idx = sorted(rnd.sample(range(n), t + k if n > 100 else rnd.randrange(t + k, n + 1)))
        sh = [fft[i] for i in idx]
        if n <= 100:
            assert [v % prime for v in pss.reconstruct_newton(idx, sh)] == s
        assert pss.reconstruct_lagrange(idx, sh) == s


def test_p62_parameters():
    """the 62-bit prime and its roots of unity."""
    assert P62.bit_length() == 62 and 4 * P62 < 2 ** 64 <= 8 * P62
    for order, w in po.P62_OMEGA.items():
        assert pow(w, order, P62) == 1
        for d in range(1, order):
            if order % d == 0:
                assert pow(w, d, P62) != 1
    for sc in load_golden("p62.json")["scenarios"]:
        modes = ["canonical"] + (["rust_signed"] if "rust_signed" in sc["stages"] else [])
        for mode in modes:
            r = po.full_aggregation(sc["aggregation"], sc["inputs"], sc.get("mask_rand", [[]] * len(sc["inputs"])),
                                    sc["share_rand"], sc["clerk_subset"], mode)
            assert r == sc["stages"][mode]
            assert r["positive"] == [sum(col) % P62 for col in zip(*sc["inputs"])]


# ---- C oracle == Python oracle -------------------------------------------------------------------------------
@pytest.mark.parametrize("mode_name,mode", [("canonical", 0), ("rust_signed", 1)])
def test_c_oracle_additive_and_combine(mode_name, mode):
    rnd = random.Random(3)
    for q, n, dim in [(433, 3, 10), (P62, 3, 17), (P62, 5, 4), (97, 1, 6), (2, 3, 9)]:
        secrets = [rnd.randrange(q) for _ in range(dim)]
        rand = [rnd.randrange(q) for _ in range(dim * (n - 1))]
        gen = po.AdditiveSecretSharing(n, q, mode_name)
        want = po.generate(gen, secrets, rand)
        got = coracle.additive_generate(q, n, secrets, rand, mode)
        assert got.tolist() == want
        rows = [[rnd.randrange(-(q - 1), q) for _ in range(dim)] for _ in range(6)]
        assert coracle.combine(q, rows, mode).tolist() == po.combine(rows, q, mode_name)
        a = [rnd.randrange(-(q - 1), q) for _ in range(dim)]
        b = [rnd.randrange(q) for _ in range(dim)]
        rem = po._rem(mode_name)
        assert coracle.addsub(a, b, q, False, mode).tolist() == [rem(x + y, q) for x, y in zip(a, b)] or mode == 0
        assert coracle.addsub(a, b, q, True, mode).tolist() == [rem(x - y, q) for x, y in zip(a, b)] or mode == 0
        assert coracle.addsub(a, b, q, True, 0).tolist() == [(x - y) % q for x, y in zip(a, b)]


@pytest.mark.parametrize("k,t,n,o2,o3", [(3, 1, 8, 8, 9), (3, 4, 8, 8, 9), (8, 2, 26, 16, 27), (8, 7, 26, 16, 27),
                                         (5, 3, 26, 16, 27)])
def test_c_oracle_packed(k, t, n, o2, o3):
    rnd = random.Random(k * 10 + t)
    w2, w3 = po.P62_OMEGA[o2], po.P62_OMEGA[o3]
    dim = 4 * k + 1
    secrets = [rnd.randrange(-(1 << 63), 1 << 63) for _ in range(dim)]
    B = (dim + k - 1) // k
    rand = [rnd.randrange(-(1 << 63), 1 << 63) for _ in range(B * t)]
    gen = po.PackedShamirGenerator(t, n, k, P62, w2, w3, "canonical")
    want = po.generate(gen, secrets, rand)
    got = coracle.packed_generate(P62, k, t, n, w2, w3, secrets, rand)
    assert got.tolist() == want
    assert coracle.packed_share_matrix(P62, k, t, n, w2, w3).tolist() == gen.pss.share_matrix()
    idx = sorted(rnd.sample(range(n), t + k + (1 if t + k < n else 0)))
    rec = po.PackedShamirReconstructor(dim, t, n, k, P62, w2, w3, "canonical")
    want_s = rec.reconstruct([(i, want[i]) for i in idx])
    assert want_s == [s % P62 for s in secrets]
    assert coracle.packed_reconstruct(P62, k, t, w2, w3, dim, idx, got[idx]).tolist() == want_s
    with pytest.raises(ValueError):
        coracle.packed_reconstruct(P62, k, t, w2, w3, dim, idx[:t + k - 1], got[idx[:t + k - 1]])


def test_c_oracle_chacha_combine():
    rnd = random.Random(9)
    for q in (433, P62, (1 << 61) + 1):
        seeds = [[rnd.getrandbits(32) for _ in range(4)] for _ in range(3)]
        m = po.ChaChaMasker(q, 50, 128, "canonical")
        assert coracle.chacha_combine(seeds, q, 50).tolist() == m.combine(seeds)
    assert coracle.chacha_combine([], 433, 4).tolist() == [0, 0, 0, 0]


def test_drbg_call_key_c_equals_python_and_golden():
    """the per-call key derivation of sda-drbg-v1 (one RFC 7539 ChaCha20 block; the block function itself is pinned by
    the RFC vectors above)"""
    g = load_golden("drbg.json")
    key = bytes.fromhex(g["key_hex"])
    for c in g["call_keys"]:
        assert po.drbg_call_key(key, c["call_index"]).hex() == c["key_hex"]
        assert coracle.drbg_call_key(key, c["call_index"]).hex() == c["key_hex"]
    assert len({c["key_hex"] for c in g["call_keys"]}) == len(g["call_keys"])


def test_drbg_spec_c_equals_python_and_golden():
    g = load_golden("drbg.json")
    key = bytes.fromhex(g["key_hex"])
    for c in g["cases"]:
        want = po.drbg_fill(key, c["stream"], c["batches"], c["T"], c["modulus"], c["rounds"])
        assert want == c["values"]
        assert coracle.drbg_fill(key, c["stream"], c["batches"], c["T"], c["modulus"], c["rounds"]).tolist() == want
        assert all(0 <= v < c["modulus"] for v in want)
    # retry stream exercised: heavy-rejection modulus
    m = (1 << 61) + 1
    assert coracle.drbg_fill(key, 3, 300, 2, m).tolist() == po.drbg_fill(key, 3, 300, 2, m)


def _np_chacha20_words(key_words, counters, stream):
    """ChaCha20 blocks of sda-drbg-v1's attempt-0 layout for an array of block counters, vectorised (test-local; the block
    function itself is pinned by the RFC 7539 vectors above): returns a (16, len) uint32 array"""
    n = len(counters)
    st = [np.full(n, c, dtype=np.uint32) for c in po.CHACHA_CONST] + [np.full(n, w, dtype=np.uint32) for w in key_words]
    st += [(counters & 0xFFFFFFFF).astype(np.uint32), (counters >> 32).astype(np.uint32),
           np.full(n, stream & 0xFFFFFFFF, dtype=np.uint32), np.full(n, (stream >> 32) & 0xFFFFFF, dtype=np.uint32)]
    x = [v.copy() for v in st]

    def rot(v, r):
        return (v << np.uint32(r)) | (v >> np.uint32(32 - r))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] = rot(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b] = rot(x[b] ^ x[c], 12)
        x[a] += x[b]; x[d] = rot(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b] = rot(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return np.stack([x[i] + st[i] for i in range(16)])


def test_drbg_paired_rule_retry_stream():
    """the PAIRED rule (moduli <= 0x7F7F7F) rejects a candidate word with probability below 2^-18, so the retry stream is found,
    not forced: every candidate word of 6 Mi pairs is scanned (vectorised), the rejected pairs located, and there the C
    restatement must equal the big-int one (which also ties the scan's word mapping to the spec: accepted neighbours compared)"""
    key = bytes(range(32))
    kw = [int.from_bytes(key[4 * j:4 * j + 4], "little") for j in range(8)]
    m, stream, B = 8355709, 12, 6 << 20                  # T = 2: one pair per batch, block counter = b >> 3
    with np.errstate(over="ignore"):
        o = _np_chacha20_words(kw, np.arange(B >> 3, dtype=np.uint64), stream)
    b = np.arange(B)
    c, e = (b & 7) >> 1, b & 1
    x = (o[8 * e + c, b >> 3].astype(np.uint64) << np.uint64(32)) | o[8 * e + 4 + c, b >> 3].astype(np.uint64)
    thr2 = (1 << 64) % (m * m)
    with np.errstate(over="ignore"):
        lo = (x * np.uint64(m)) * np.uint64(m)           # lo64(x m m) = lo64(lo64(x m) m)
    rejected = np.nonzero(lo < np.uint64(thr2))[0]
    assert 1 <= len(rejected) <= 40, len(rejected)
    got = coracle.drbg_fill(key, stream, B, 2, m).reshape(B, 2)
    for bb in rejected.tolist() + [0, 1, 7, 8, B - 1]:
        assert got[bb].tolist() == [po.drbg_value(key, stream, bb, 2, 0, m), po.drbg_value(key, stream, bb, 2, 1, m)], bb
    assert got.min() >= 0 and got.max() < m
    # and the first-attempt values of the accepted pairs are the scan's own
    acc = np.ones(B, dtype=bool); acc[rejected] = False
    ra = np.array([(int(v) * m) >> 64 for v in x[:4096]])
    assert np.array_equal(got[:4096, 0][acc[:4096]], ra[acc[:4096]])


def test_csprng_share_map_c_equals_python_and_golden():
    """the library's systematic CSPRNG share map (include/sda_hip.h): the C restatement and the big-int one against the
    committed cases, draws = shares 0..t-1, and the tie to tss's own map - share(secrets, implied randomness) is the same
    sharing, through the matrix form AND (tss-valid small prime) tss's FFT form."""
    g = load_golden("drbg.json")
    key = bytes.fromhex(g["key_hex"])
    assert len(g["share_map_cases"]) >= 5
    for c in g["share_map_cases"]:
        k, t, n, m = c["secret_count"], c["privacy_threshold"], c["share_count"], c["modulus"]
        w2, w3 = c["omega_secrets"], c["omega_shares"]
        B = -(-len(c["secrets"]) // k)
        assert po.drbg_fill(key, c["stream"], B, t, m) == c["draws"]
        got, implied = coracle.packed_generate_systematic(m, k, t, n, w2, w3, c["secrets"], c["draws"], want_implied=True)
        assert got.tolist() == c["systematic_shares"] and implied.tolist() == c["implied_tss_randomness"]
        assert got[:t].T.reshape(-1).tolist() == c["draws"]
        assert coracle.packed_generate(m, k, t, n, w2, w3, c["secrets"], implied).tolist() == c["systematic_shares"]
        assert coracle.packed_generate_csprng(m, k, t, n, w2, w3, c["secrets"], c["draws"], 1).tolist() == c["systematic_shares"]
        pss = po.PackedSecretSharing(t, n, k, m, w2, w3)
        for b in range(B):
            batch = c["secrets"][b * k:(b + 1) * k]
            batch = batch + [0] * (k - len(batch))
            col = [row[b] for row in c["systematic_shares"]]
            assert pss.share_systematic(batch, c["draws"][b * t:(b + 1) * t]) == col
            if m * m < 2 ** 62 and pss.is_fft_shape():                   # tss's own transform form, canonicalised
                fft = pss.share_fft(batch, c["implied_tss_randomness"][b * t:(b + 1) * t], "rust_signed")
                assert [x % m for x in fft] == col
            # any t + k of the shares give the secrets back (first-t direct rows included)
            idx = list(range(t + k))
            assert pss.reconstruct_lagrange(idx, [col[i] for i in idx]) == batch


def test_synthetic_input_and_baseline_pass():
    a = coracle.fill_synthetic(2, 5, 10, 0x5DA5DA5DA5DA5DA5, P62)
    assert a.tolist() == [[po.synthetic_secret(0x5DA5DA5DA5DA5DA5, 10 + p, i, P62) for i in range(5)] for p in range(2)]
    key = bytes(range(32))
    for packed, n, k, t, o2, o3 in [(1, 8, 3, 1, 8, 9), (0, 3, 1, 2, 8, 9)]:
        done, sums = coracle.baseline_pass(packed, P62, n, k, t, po.P62_OMEGA[o2], po.P62_OMEGA[o3], 4, 50, 0, 1, key)
        assert done == 200
        sec = coracle.fill_synthetic(4, 50, 0, 1, P62)
        if packed:
            rec = coracle.packed_reconstruct(P62, k, t, po.P62_OMEGA[o2], po.P62_OMEGA[o3], 50, [1, 3, 5, 7], sums[[1, 3, 5, 7]])
        else:
            rec = coracle.combine(P62, sums)
        assert rec.tolist() == coracle.combine(P62, sec).tolist()


def test_positive_and_errors():
    assert po.positive([-1, 0, 5], 433) == [432, 0, 5] == coracle.positive([-1, 0, 5], 433).tolist()
    with pytest.raises(ValueError, match="Wrong dimension"):
        po.combine([[1, 2], [1]], 433)
    with pytest.raises(ValueError, match="Mismatching dimension"):
        po.AdditiveSecretSharing(3, 433).reconstruct([(0, [1, 2]), (1, [1])])
    with pytest.raises(ValueError, match="Not enough shares"):
        po.PackedShamirReconstructor(3, 4, 8, 3, 433, 354, 150).reconstruct([(i, [1]) for i in range(6)])
    with pytest.raises(ValueError, match="Batch input wrong length"):
        po.AdditiveSecretSharing(3, 433).generate_for_batch([1, 2], [0, 0])
    assert po.combine([], 433) == [] and po.AdditiveSecretSharing(3, 433).reconstruct([]) == []
    assert po.trunc_rem(-394, 433) == -394 and po.trunc_rem(1 - 400, 433) == -399 and po.trunc_rem(-866, 433) == 0


def test_varint_codec_oracle():
    """integer-encoding 1.0 VarInt for i64 (sodium.rs:39,86): published zig-zag/LEB128 vectors + C == Python."""
    assert po.varint_encode_i64(0) == b"\x00" and po.varint_encode_i64(-1) == b"\x01"
    assert po.varint_encode_i64(1) == b"\x02" and po.varint_encode_i64(-2) == b"\x03"
    assert po.varint_encode_i64(75) == bytes([0x96, 0x01])                      # zigzag(75) = 150: the LEB128 textbook case
    assert po.varint_encode_i64(2147483647) == bytes([0xfe, 0xff, 0xff, 0xff, 0x0f])       # protobuf sint vectors
    assert po.varint_encode_i64(-2147483648) == bytes([0xff, 0xff, 0xff, 0xff, 0x0f])
    assert po.varint_encode_i64(2 ** 63 - 1) == bytes([0xfe] + [0xff] * 8 + [0x01])
    assert po.varint_encode_i64(-2 ** 63) == bytes([0xff] * 9 + [0x01])
    rnd = random.Random(4)
    vals = [rnd.choice([0, 1, -1, 63, 64, -64, -65, P62 - 1, -(1 << 63), (1 << 63) - 1, rnd.randrange(-(1 << 63), 1 << 63),
                        rnd.randrange(-1000, 1000)]) for _ in range(4000)]
    enc = po.varint_encode(vals)
    assert po.varint_decode(enc) == vals
    assert coracle.varint_encode(vals) == enc and coracle.varint_decode(enc).tolist() == vals
    assert po.varint_decode(bytes([0x02, 0x80])) == [1, 0]                      # unterminated tail: a partial value
    assert coracle.varint_decode(bytes([0x02, 0x80])).tolist() == [1, 0]


# ---- algebraic properties under hypothesis ------------------------------------------------------
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=60, deadline=None)
@given(st.integers(min_value=0, max_value=2 ** 32), st.integers(min_value=1, max_value=4), st.integers(min_value=1, max_value=9),
       st.sampled_from(["None", "Full", "ChaCha"]), st.booleans())
def test_linear_invariant_property(seed, participants, dim, mask_kind, packed):
    """reveal(aggregate) == sum of inputs mod q for every masking scheme x sharing scheme, every
    >= (t+k)-subset of clerks, any randomness, in both value modes - the invariant the reference's
    full_loop tests assert on one fixed input."""
    rnd = random.Random(seed)
    q = 433
    if packed:
        shr = dict(po.PSS_433)
        k, t, n = 3, 4, 8
        subset = sorted(rnd.sample(range(n), rnd.randrange(t + k, n + 1)))
    else:
        n = rnd.randrange(1, 5)
        shr = dict(kind="Additive", share_count=n, modulus=q)
        k, t, subset = 1, n - 1, None
    msk = {"None": dict(kind="None"), "Full": dict(kind="Full", modulus=q),
           "ChaCha": dict(kind="ChaCha", modulus=q, dimension=dim, seed_bitsize=rnd.choice([32, 64, 128, 256]))}[mask_kind]
    a = dict(vector_dimension=dim, modulus=q, masking_scheme=msk, committee_sharing_scheme=shr)
    inputs = [[rnd.randrange(-500, 1000) for _ in range(dim)] for _ in range(participants)]
    nb = (dim + k - 1) // k
    mr = [([rnd.randrange(q) for _ in range(dim)] if mask_kind == "Full" else
           [rnd.getrandbits(32) for _ in range((msk.get("seed_bitsize", 0) + 31) // 32)]) for _ in inputs]
    sr = [[rnd.randrange(q - 1) for _ in range(nb * t)] for _ in inputs]
    want = [sum(col) % q for col in zip(*inputs)]
    for mode in ("rust_signed", "canonical"):
        r = po.full_aggregation(a, inputs, mr, sr, subset, mode)
        assert r["positive"] == want
        assert [v % q for v in r["output"]] == want


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=-(2 ** 63), max_value=2 ** 63 - 1), min_size=0, max_size=200))
def test_varint_roundtrip_property(values):
    enc = po.varint_encode(values)
    assert po.varint_decode(enc) == values
    assert coracle.varint_encode(values) == enc
    assert len(enc) == sum(max(1, ((((v << 1) ^ (v >> 63)) & po.MASK64).bit_length() + 6) // 7) for v in values)


def test_c_oracle_under_sanitizers():
    """the host oracle is built with -fsanitize=address,undefined and run over the reference's end-to-end
    vectors, the mask paths, the DRBG and the wire codec with exact-size buffers (oracle/selftest.c)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-s", "check-sanitize"], capture_output=True, text=True,
                       timeout=300)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr or "cannot find -lubsan" in r.stderr):
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0 and "oracle selftest: OK" in r.stdout, r.stdout + r.stderr


# ---- reference-generated fixtures (tests/reference_harness: the real crates, run where cargo exists) -----------------------
REFERENCE_GENERATED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_generated.json")


def _load_harness_generator():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_harness", "gen_inputs.py")
    spec = importlib.util.spec_from_file_location("reference_harness_gen_inputs", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_reference_generated(doc, gen):
    """What tests/reference_harness printed (threshold-secret-sharing 0.2, rand 0.3) against the oracle:
      * ChaChaRng::from_seed + gen_range(0_i64, q) prefixes, first next_u64 / next_u32s == the oracle's restatement
        (chacha.rs:36-39: word order of next_u64, zone rejection, seeds shorter / longer than four words);
      * oracle-reconstruct(shares made by tss) == the secrets, tss-reconstruct(shares made by the oracle) == the secrets,
        for arbitrary clerk subsets (packed_shamir.rs:42,76: the evaluation-point convention index i <-> omega_shares^(i+1))."""
    assert doc["provenance"].startswith("reference-generated")
    by_name = {c["name"]: c for c in doc["chacha"]}
    for name, seed, q, count in gen.CHACHA_CASES:
        c = by_name[name]
        assert c["seed"] == seed and c["modulus"] == q
        rng = po.ChaChaRng(seed)
        assert c["masks"] == [rng.gen_range_i64(0, q) for _ in range(count)], name
        assert c["first_next_u64"] == po.ChaChaRng(seed).next_u64(), name
        r = po.ChaChaRng(seed)
        assert c["first_next_u32s"] == [r.next_u32() for _ in range(4)], name
    by_name = {c["name"]: c for c in doc["raw"]}
    for name, seed, lo, hi, words in gen.RAW_CASES:
        c = by_name[name]
        assert c["seed"] == seed and c["counter_low"] == lo and c["counter_high"] == hi
        r = po.ChaChaRng(seed)
        if lo or hi:
            r.set_counter(lo, hi)
        assert c["next_u32s"] == [r.next_u32() for _ in range(words)], name
    by_name = {c["name"]: c for c in doc["pss"]}
    for args in gen.PSS_CASES:
        want = gen.pss_case(*args)
        c = by_name[want["name"]]
        p, n = want["p"], want["n"]
        pss = po.PackedSecretSharing(want["t"], n, want["k"], p, want["w2"], want["w3"])
        assert c["secrets"] == want["secrets"] and c["subset"] == want["subset"] and c["reconstruct_limit"] == pss.reconstruct_limit()
        # tss reconstructed the ORACLE's shares
        assert [v % p for v in c["tss_reconstruct_of_oracle_shares"]] == want["secrets"], want["name"]
        assert [v % p for v in c["tss_reconstruct_of_tss_shares"]] == want["secrets"], want["name"]
        # the oracle reconstructs TSS's shares: from every clerk, and from the subset (Newton path and Lagrange-matrix path)
        tss_shares = [v % p for v in c["tss_shares"]]
        assert len(tss_shares) == n
        every = list(range(n))
        assert [v % p for v in pss.reconstruct(every, tss_shares, "canonical")] == want["secrets"], want["name"]
        sub = want["subset"]
        assert [v % p for v in pss.reconstruct(sub, [tss_shares[i] for i in sub], "canonical")] == want["secrets"]
        assert [v % p for v in pss.reconstruct_lagrange(sub, [tss_shares[i] for i in sub])] == want["secrets"]


def _simulated_reference_output(gen):
    """the document the harness WOULD print if the crates behave as the oracle restates them (used to keep the consumer
    above exercised while no machine with cargo has produced the real file)"""
    import random as _r
    doc = {"provenance": "reference-generated (SIMULATED by the oracle for the consumer's self-test)", "chacha": [], "raw": [], "pss": []}
    for name, seed, lo, hi, words in gen.RAW_CASES:
        r = po.ChaChaRng(seed)
        if lo or hi:
            r.set_counter(lo, hi)
        doc["raw"].append({"name": name, "seed": seed, "counter_low": lo, "counter_high": hi, "next_u32s": [r.next_u32() for _ in range(words)]})
    for name, seed, q, count in gen.CHACHA_CASES:
        rng = po.ChaChaRng(seed)
        r = po.ChaChaRng(seed)
        doc["chacha"].append({"name": name, "seed": seed, "modulus": q, "masks": [rng.gen_range_i64(0, q) for _ in range(count)],
                              "first_next_u64": po.ChaChaRng(seed).next_u64(), "first_next_u32s": [r.next_u32() for _ in range(4)]})
    rnd = _r.Random(99)
    for args in gen.PSS_CASES:
        w = gen.pss_case(*args)
        pss = po.PackedSecretSharing(w["t"], w["n"], w["k"], w["p"], w["w2"], w["w3"])
        fresh = [rnd.randrange(w["p"] - 1) for _ in range(w["t"])]                       # tss draws its own randomness
        shares = pss.share(w["secrets"], fresh, "rust_signed")                          # signed, as tss returns them
        doc["pss"].append({"name": w["name"], "secrets": w["secrets"], "subset": w["subset"], "reconstruct_limit": pss.reconstruct_limit(),
                           "tss_shares": shares,
                           "tss_reconstruct_of_oracle_shares": pss.reconstruct(w["subset"], [w["shares"][i] for i in w["subset"]], "rust_signed"),
                           "tss_reconstruct_of_tss_shares": pss.reconstruct(list(range(w["n"])), shares, "rust_signed")})
    return doc


def test_reference_harness_inputs_are_current():
    """tests/reference_harness/src/oracle_inputs.rs (what the Rust harness feeds the real crates) is what the oracle
    generates today; and the consumer of the harness's output accepts a faithful document and refuses a wrong one"""
    gen = _load_harness_generator()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_harness", "src", "oracle_inputs.rs")
    assert open(path).read() == gen.render(), "run python tests/reference_harness/gen_inputs.py"
    doc = _simulated_reference_output(gen)
    check_reference_generated(doc, gen)
    bad = json.loads(json.dumps(doc))
    bad["chacha"][5]["masks"][7] ^= 1
    with pytest.raises(AssertionError):
        check_reference_generated(bad, gen)
    bad = json.loads(json.dumps(doc))
    bad["raw"][1]["next_u32s"][16:32] = doc["raw"][0]["next_u32s"][16:32]             # the block after the carry computed WITHOUT the carry
    with pytest.raises(AssertionError):
        check_reference_generated(bad, gen)
    bad = json.loads(json.dumps(doc))
    bad["pss"][0]["tss_shares"] = bad["pss"][0]["tss_shares"][1:] + bad["pss"][0]["tss_shares"][:1]     # shares off by one clerk index
    with pytest.raises(AssertionError):
        check_reference_generated(bad, gen)


def test_reference_generated_fixtures():
    """Consumes tests/golden/reference_generated.json - the output of tests/reference_harness (the crates the reference
    links, run on a machine with cargo).  ABSENT in this tree: the image has no rustc / cargo and no network, so share-level
    parity still rests on recalled crate vectors - this test says so loudly instead of passing."""
    if not os.path.exists(REFERENCE_GENERATED):
        pytest.skip("PARITY PIN MISSING: tests/golden/reference_generated.json does not exist - nobody has run "
                    "`cd tests/reference_harness && cargo run --release > ../golden/reference_generated.json` yet "
                    "(needs cargo + crates.io); share-level parity rests on RECALLED tss 0.2 / rand 0.3 vectors until then")
    check_reference_generated(json.load(open(REFERENCE_GENERATED)), _load_harness_generator())


def test_every_golden_file_states_its_provenance():
    """each fixture file says, part by part, which of reference-held / recalled / published-RFC / oracle-generated pins it"""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    words = ("reference-held", "recalled", "published-RFC", "oracle-generated", "reference-generated", "published")
    files = sorted(f for f in os.listdir(golden) if f.endswith(".json"))
    assert {"kats.json", "full_loop.json", "p62.json", "drbg.json", "sealedbox.json"} <= set(files)
    for f in files:
        prov = json.load(open(os.path.join(golden, f))).get("provenance")
        assert prov, f"{f} has no provenance key"
        texts = [prov] if isinstance(prov, str) else list(prov.values())
        assert all(any(w in t for w in words) for t in texts), (f, texts)
