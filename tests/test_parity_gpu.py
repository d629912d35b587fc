"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on the same
inputs.  Bit-exact (integer work).  Run on the GPU box with  pytest -m gpu."""
import os

import numpy as np
import pytest

from conftest import load_golden, set_knob


def _free_port():
    """a rendezvous port nobody holds right now (fixed ports collide when GPU tests run side by side)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])

pytestmark = pytest.mark.gpu

P62 = 4611686006577364993
W = {8: 631229665360524489, 9: 3451275676410824977, 16: 2589100645267092065, 27: 365137883145458390,
     32: 1942624553499164220, 64: 2724396144719537715, 128: 4438016560451165920, 3: 3}   # 32 / 64 / 128: elements of that order; 3: of large order
KEY = bytes(range(32))


def _scheme(d):
    from sda_amd import crypto
    if d["kind"] == "Additive":
        return crypto.Additive(d["share_count"], d["modulus"])
    return crypto.PackedShamir(d["secret_count"], d["share_count"], d["privacy_threshold"], d["prime_modulus"],
                               d["omega_secrets"], d["omega_shares"])


def _mask_scheme(d):
    from sda_amd import crypto
    if d["kind"] == "None":
        return crypto.NoMask()
    if d["kind"] == "Full":
        return crypto.Full(d["modulus"])
    return crypto.ChaCha(d["modulus"], d["dimension"], d["seed_bitsize"])


def _run_scenario(sc):
    from sda_amd import crypto
    a = sc["aggregation"]
    agg = crypto.Aggregation(a["vector_dimension"], a["modulus"], _mask_scheme(a["masking_scheme"]),
                             _scheme(a["committee_sharing_scheme"]))
    mask_rand = sc.get("mask_rand")
    if a["masking_scheme"]["kind"] == "None":
        mask_rand = None
    got = crypto.full_aggregation(agg, sc["inputs"], mask_rand, sc["share_rand"], sc["clerk_subset"])
    want = sc["stages"]["canonical"]
    q = a["committee_sharing_scheme"].get("modulus", a["committee_sharing_scheme"].get("prime_modulus"))
    for p in range(len(sc["inputs"])):
        assert [list(map(int, row)) for row in got["shares"][p]] == want["shares"][p], f"shares of participant {p}"
        assert list(map(int, got["masked"][p])) == want["masked"][p]
        if a["masking_scheme"]["kind"] != "None":
            assert list(map(int, got["masks"][p])) == want["masks"][p]
    assert [list(map(int, s)) for s in got["clerk_sums"]] == want["clerk_sums"]
    assert list(map(int, got["combined_mask"])) == want["combined_mask"]
    assert list(map(int, got["masked_output"])) == want["masked_output"]
    assert list(map(int, got["output"])) == want["output"]
    assert list(map(int, got["positive"])) == want["positive"]
    if "expected_positive" in sc:
        assert list(map(int, got["positive"])) == sc["expected_positive"]
    # rust_signed stage values agree modulo q with what the GPU emits
    if "rust_signed" in sc["stages"]:
        rs = sc["stages"]["rust_signed"]
        assert [[v % q for v in s] for s in rs["clerk_sums"]] == [list(map(int, s)) for s in got["clerk_sums"]]
        assert rs["positive"] == list(map(int, got["positive"]))


@pytest.mark.parametrize("idx", range(7))
def test_reference_full_loop_vectors(gpu, idx):
    """F0-F4: the reference's own end-to-end vectors (full_loop.rs:29-67,148; README.md:157)."""
    _run_scenario(load_golden("full_loop.json")["scenarios"][idx])


def _run_scenario_rust_signed(sc):
    """SDA_VALUES_RUST_SIGNED: every stage BIT FOR BIT equal to the oracle's `rust_signed` stages - the reference's own
    representatives (Rust's truncated `%`), for the paths whose arithmetic is in /root/reference
    (additive.rs, combiner.rs, full.rs, chacha.rs).  Packed Shamir's shares / reconstruction stay canonical (tss's signed
    values are an un-vendored crate's): there the comparison is modulo the prime, as before."""
    from sda_amd import crypto
    a = sc["aggregation"]
    agg = crypto.Aggregation(a["vector_dimension"], a["modulus"], _mask_scheme(a["masking_scheme"]),
                             _scheme(a["committee_sharing_scheme"]))
    mask_rand = sc.get("mask_rand")
    if a["masking_scheme"]["kind"] == "None":
        mask_rand = None
    got = crypto.full_aggregation(agg, sc["inputs"], mask_rand, sc["share_rand"], sc["clerk_subset"], value_mode=crypto.RUST_SIGNED)
    want = sc["stages"]["rust_signed"]
    additive = a["committee_sharing_scheme"]["kind"] == "Additive"
    q = a["committee_sharing_scheme"].get("modulus", a["committee_sharing_scheme"].get("prime_modulus"))
    same = (lambda g, w: list(map(int, g)) == list(w)) if additive else (lambda g, w: [int(v) % q for v in g] == [v % q for v in w])
    for p in range(len(sc["inputs"])):
        assert list(map(int, got["masked"][p])) == want["masked"][p], f"masked secrets of participant {p}"        # exact, both schemes
        if a["masking_scheme"]["kind"] != "None":
            assert list(map(int, got["masks"][p])) == want["masks"][p]
        for c, row in enumerate(got["shares"][p]):
            assert same(row, want["shares"][p][c]), f"share {c} of participant {p}"
    for c, row in enumerate(got["clerk_sums"]):
        assert same(row, want["clerk_sums"][c]), f"clerk sum {c}"
    assert list(map(int, got["combined_mask"])) == want["combined_mask"]
    assert same(got["masked_output"], want["masked_output"])
    assert same(got["output"], want["output"])
    assert list(map(int, got["positive"])) == want["positive"] == sc.get("expected_positive", want["positive"])
    return additive


@pytest.mark.parametrize("name,idx", [("full_loop.json", i) for i in range(7)] + [("p62.json", i) for i in range(7)])
def test_reference_signed_representatives_bit_for_bit(gpu, name, idx):
    """F0-F4 (the reference's own scenarios) and the 62-bit scenarios in the reference's OWN value representation"""
    sc = load_golden(name)["scenarios"][idx]
    if "rust_signed" not in sc["stages"]:
        pytest.skip("this scenario has no rust_signed stages (the reference's packed arithmetic overflows at this prime)")
    _run_scenario_rust_signed(sc)


@pytest.mark.parametrize("q", [433, P62, (1 << 62) - 57, 2])
def test_signed_mode_random_any_i64_vs_oracle(gpu, q):
    """the signed kernels against the oracle's rust_signed restatement on ANY i64 inputs (the reference never range-checks;
    the oracle forms every sum exactly): additive generate with injected draws, combine over many participants (signs
    depend on the history), additive reconstruct, full mask / unmask, ChaCha mask / unmask"""
    from sda_amd import crypto
    from oracle import pyoracle as po
    rng = np.random.default_rng(q % 1000)
    n, dim, P = 4, 257, 37
    big = lambda size: rng.integers(-(1 << 62) + 1, (1 << 62) - 1, size=size, dtype=np.int64)
    small = lambda size: rng.integers(-q + 1, q, size=size, dtype=np.int64) if q < (1 << 62) else big(size)
    sch = crypto.Additive(n, q)
    gen = crypto.ShareGenerator(sch).set_value_mode("rust_signed")
    osch = po.AdditiveSecretSharing(n, q, "rust_signed")
    rows = []
    for pick in (big, small):
        secrets, rand = pick(dim), pick(dim * (n - 1))
        got = gen.generate(secrets, rand)
        want = po.generate(osch, [int(v) for v in secrets], [int(v) for v in rand])
        assert [list(map(int, r)) for r in got] == want
        rows.append(got)
    comb = crypto.ShareCombiner(sch).set_value_mode("rust_signed")
    vecs = [small(dim) for _ in range(P)] + [big(dim) for _ in range(3)]
    want = po.combine([[int(v) for v in r] for r in vecs], q, "rust_signed")
    assert list(map(int, comb.combine(vecs))) == want
    comb.begin(dim)                                                   # the streaming form keeps the same state
    comb.update(np.stack(vecs[:20])); comb.update(np.stack(vecs[20:]))
    assert list(map(int, comb.finish())) == want
    rec = crypto.SecretReconstructor(sch, dim).set_value_mode("rust_signed")
    clerk = [small(dim) for _ in range(n)]
    assert list(map(int, rec.reconstruct(list(enumerate(clerk))))) == osch.reconstruct([(i, [int(v) for v in r]) for i, r in enumerate(clerk)])
    # masks
    fm = po.FullMasker(q, "rust_signed")
    m = crypto.SecretMasker(crypto.Full(q)).set_value_mode("rust_signed")
    secrets, draws = big(dim), small(dim)
    mask, masked = m.mask(secrets, draws)
    wmask, wmasked = fm.mask([int(v) for v in secrets], [int(v) for v in draws])
    assert list(map(int, mask)) == wmask and list(map(int, masked)) == wmasked
    mc = crypto.MaskCombiner(crypto.Full(q)).set_value_mode("rust_signed")
    assert list(map(int, mc.combine(vecs))) == fm.combine([[int(v) for v in r] for r in vecs])
    um = crypto.SecretUnmasker(crypto.Full(q)).set_value_mode("rust_signed")
    a_, b_ = big(dim), big(dim)
    assert list(map(int, um.unmask((a_, b_)))) == fm.unmask([int(v) for v in a_], [int(v) for v in b_])
    if q > 2:
        cm = po.ChaChaMasker(q, dim, 128, "rust_signed")
        m2 = crypto.SecretMasker(crypto.ChaCha(q, dim, 128)).set_value_mode("rust_signed")
        seed = [1, 2, 3, 0xFFFFFFFF]
        mask, masked = m2.mask(secrets, seed)
        wmask, wmasked = cm.mask([int(v) for v in secrets], seed)
        assert list(map(int, mask)) == wmask and list(map(int, masked)) == wmasked
    # what the mode does not cover says so
    from sda_amd import capi
    with pytest.raises(capi.SdaError) as e:
        crypto.ShareGenerator(crypto.PackedShamir(3, 8, 1, P62, W[8], W[9])).set_value_mode("rust_signed")
    assert e.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.parametrize("idx", range(7))
def test_p62_vectors(gpu, idx):
    """62-bit prime configurations (BASELINE configs 2-4 shapes) incl. un-range-checked i64 inputs."""
    _run_scenario(load_golden("p62.json")["scenarios"][idx])


def test_kats_share_and_reconstruct(gpu):
    from sda_amd import crypto
    k = load_golden("kats.json")["kats"]
    sch = _scheme(k["B3_share"]["scheme"])
    g = crypto.ShareGenerator(sch)
    shares = g.generate(k["B3_share"]["secrets"], k["B3_share"]["randomness"])
    assert [int(s[0]) for s in shares] == k["B3_share"]["expected"]
    r = crypto.SecretReconstructor(sch, 3)
    for idx in k["B5_reconstruct"]["index_sets"]:
        out = r.reconstruct([(i, [k["B5_reconstruct"]["shares"][i]]) for i in idx])
        assert list(map(int, out)) == k["B5_reconstruct"]["expected"]


# ---- share generation vs the C oracle ---------------------------------------------------------------
@pytest.mark.parametrize("n,q,dim", [(3, 433, 10), (3, P62, 1), (2, P62, 2), (5, P62, 1001), (1, 97, 33),
                                     (3, (1 << 61) + 20, 4096), (3, 2, 17)])
def test_additive_generate_injected(gpu, n, q, dim):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(n * 1000 + dim)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    rand = rng.integers(-(1 << 62), 1 << 62, size=dim * (n - 1), dtype=np.int64)
    got = crypto.ShareGenerator(crypto.Additive(n, q)).generate(secrets, rand)
    want = coracle.additive_generate(q, n, secrets, rand, mode=0)
    assert np.array_equal(got, want)
    # and mod q against the reference's signed arithmetic on in-range inputs
    s2 = np.mod(secrets, q)
    r2 = np.mod(rand, q)
    signed = coracle.additive_generate(q, n, s2, r2, mode=1)
    assert np.array_equal(np.mod(signed, q), crypto.ShareGenerator(crypto.Additive(n, q)).generate(s2, r2))


PACKED_SHAPES = [(3, 1, 8, 8, 9), (3, 4, 8, 8, 9), (8, 2, 26, 16, 27), (8, 7, 26, 16, 27), (1, 1, 8, 8, 9),
                 (2, 1, 8, 8, 9), (1, 2, 8, 8, 9), (2, 5, 8, 8, 9), (4, 3, 26, 8, 27),
                 (5, 3, 26, 16, 27), (3, 0, 8, 8, 9),
                 (5, 2, 8, 8, 9), (6, 1, 8, 8, 9), (1, 6, 8, 8, 9), (7, 0, 8, 8, 9),      # the rest of the k + t = 7 family
                 (12, 3, 26, 16, 27), (10, 5, 26, 16, 27), (4, 11, 26, 16, 27),          # splits of k + t = 15
                 (6, 2, 8, 16, 9), (9, 6, 26, 16, 27), (2, 9, 26, 16, 27), (1, 4, 26, 8, 27),  # run-time (k, t) kernel
                 (10, 7, 26, 32, 27), (3, 2, 100, 8, 3), (3, 4, 80, 8, 3), (20, 11, 40, 32, 3),   # run-time (k, t), matrix in global memory
                 (20, 13, 80, 64, 3), (40, 23, 80, 64, 3),                                # the same with k + t up to 64 (one wave per SIMD)
                 (50, 20, 80, 128, 3)]                                                    # generic kernel (k + t > 64, not a tss shape)


@pytest.mark.parametrize("k,t,n,o2,o3", PACKED_SHAPES)
@pytest.mark.parametrize("dim", [1, 7, 1000, 6151])
def test_packed_generate_injected(gpu, k, t, n, o2, o3, dim):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(k * 100 + t * 10 + dim)
    secrets = rng.integers(0, P62, size=dim, dtype=np.int64)
    secrets[::5] = rng.integers(-(1 << 63), (1 << 63) - 1, size=secrets[::5].size, dtype=np.int64)
    B = (dim + k - 1) // k
    rand = rng.integers(-(1 << 63), (1 << 63) - 1, size=B * t, dtype=np.int64)
    sch = crypto.PackedShamir(k, n, t, P62, W[o2], W[o3])
    got = crypto.ShareGenerator(sch).generate(secrets, rand)
    want = coracle.packed_generate(P62, k, t, n, W[o2], W[o3], secrets, rand)
    assert got.shape == (n, B)
    assert np.array_equal(got, want)


def test_packed_generic_path_equals_fast_path(gpu, monkeypatch):
    from sda_amd import crypto
    sch = crypto.PackedShamir(3, 8, 1, P62, W[8], W[9])
    rng = np.random.default_rng(5)
    secrets = rng.integers(0, P62, size=5000, dtype=np.int64)
    fast = crypto.ShareGenerator(sch)
    fast.set_drbg_key(KEY)
    a = fast.generate(secrets)
    set_knob("SDA_FORCE_GENERIC", "1")
    slow = crypto.ShareGenerator(sch)
    slow.set_drbg_key(KEY)
    b = slow.generate(secrets)
    assert np.array_equal(a, b)


TSS_P1, TSS_P2 = 746497, 5038849                      # tss's shipped parameter sets [recalled, SURVEY.md App. B]: orders verified below


def _root(p, order):
    g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    return pow(g, (p - 1) // order, p)


@pytest.mark.parametrize("p,k,t,n,w2,w3,dim,force", [
    (TSS_P1, 100, 155, 728, 95660, 610121, 100 * 29 + 37, False),        # PSS_155_728_100: groups of 8 batches + a ragged group
    (P62, 100, 155, 728, None, None, 100 * 17 + 1, False),               # the same shape over the 62-bit prime
    (P62, 40, 23, 242, None, None, 40 * 40, False),                      # k + t + 1 = 64, n + 1 = 243
    (P62, 70, 57, 242, None, None, 70 * 19 + 3, False),                  # 128 of 243: first-level butterflies in two of three groups, odd radix-2 count
    (P62, 100, 155, 2186, None, None, 100 * 3 + 7, False),               # n + 1 = 3^7: one batch per workgroup with the twiddles in LDS, single level + two radix-9 passes
    (TSS_P2, 100, 155, 19682, 4318906, 1814687, 250, False),             # PSS_155_19682_100: one batch fills the LDS (G = 1)
    (4611686018374987777, 100, 155, 19682, None, None, 130, False),      # the same shape over the largest 62-bit prime with roots of order 256 and 3^9
    (P62, 3, 4, 8, W[8], W[9], 1000, True), (P62, 8, 7, 26, W[16], W[27], 6151, True),   # small tss-valid shapes, forced
    (433, 3, 4, 8, 354, 150, 7, True)])
def test_transform_share_generation_vs_oracle(gpu, monkeypatch, p, k, t, n, w2, w3, dim, force):
    """tss's own algorithm on the device (radix-2 inverse transform, zero-extension, radix-3 forward transform;
    packed_shamir.rs:42 -> tss share) for the large tss-valid shapes the matrix kernels do not cover, bit-exact against
    the oracle's matrix form: injected randomness, any-i64 secrets, and the device CSPRNG streams (sda-drbg-v1)."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    if force:
        set_knob("SDA_FORCE_FFT", "1")
    w2 = w2 or _root(p, k + t + 1)
    w3 = w3 or _root(p, n + 1)
    assert pow(w2, k + t + 1, p) == 1 and pow(w2, (k + t + 1) // 2, p) != 1
    assert pow(w3, n + 1, p) == 1 and pow(w3, (n + 1) // 3, p) != 1
    rng = np.random.default_rng(k * 7 + n)
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    B = gen.batch_count(dim)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    rand = rng.integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
    got = gen.generate(secrets, rand)
    want = coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)
    assert np.array_equal(got, want)
    # device CSPRNG: two participants, streams 70 and 71
    gen.set_drbg_key(KEY)
    P = 2
    sec2 = rng.integers(0, p, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec2)
    Bs = B + (B & 1)
    d_out = DeviceBuffer(P * n * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=70)
    out = d_out.to_numpy().reshape(P, n, Bs)
    for q in range(P):
        w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, 70 + q, B, t, p), gen.csprng_share_map())
        assert np.array_equal(out[q, :, :B], w), f"participant {q}"
    # and the round trip: reconstruct from t + k clerks == the secrets
    idx = sorted(rng.choice(n, size=t + k, replace=False).tolist())
    rec = crypto.SecretReconstructor(sch, dim).reconstruct([(i, out[0, i, :B]) for i in idx])
    assert np.array_equal(rec, sec2[0])


@pytest.mark.parametrize("a,b,split", [(2, 2, 0.5), (3, 2, 0.4), (3, 3, 0.9), (4, 3, 0.1), (4, 4, 0.5), (5, 4, 0.3), (6, 4, 0.7), (6, 5, 0.5),
                                       (7, 5, 0.2), (7, 6, 0.6), (8, 6, 0.39), (9, 6, 0.5), (9, 7, 0.5), (10, 7, 0.25)])
def test_transform_kernel_over_the_shape_space(gpu, monkeypatch, a, b, split):
    """the transform kernel for EVERY structure it can take: k + t + 1 = 2^a with a = 2..10 (odd and even: single radix-2
    level + radix-4 passes), n + 1 = 3^b with b = 2..7 (0..5 levels after the folded two: single level, radix-9 passes),
    every zero-extension pattern of the folded levels (2^a below 3^b / 9, between, above 2 * 3^b / 3), 8 batches or one per
    workgroup, twiddles in LDS or in global memory - forced for the small shapes - against the oracle's matrix form with
    injected randomness and ragged batches"""
    from sda_amd import crypto
    from oracle import coracle
    m2, m3 = 1 << a, 3 ** b
    assert m3 > m2
    kt = m2 - 1
    k = max(1, min(kt, int(round(kt * split))))
    t, n = kt - k, m3 - 1
    set_knob("SDA_FORCE_FFT", "1")
    w2, w3 = _root(P62, m2), _root(P62, m3)
    rng = np.random.default_rng(a * 100 + b)
    sch = crypto.PackedShamir(k, n, t, P62, w2, w3)
    gen = crypto.ShareGenerator(sch)
    batches = 19 if m3 <= 729 else 3
    dim = k * batches - (k // 2)                                    # ragged last batch
    B = gen.batch_count(dim)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    rand = rng.integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
    got = gen.generate(secrets, rand)
    assert np.array_equal(got, coracle.packed_generate(P62, k, t, n, w2, w3, secrets, rand))


@pytest.mark.parametrize("dim", [1, 39, 40, 40 * 7 + 3, 40 * 8, 40 * 9 - 1, 40 * 16 + 5])
def test_transform_group_boundaries(gpu, dim):
    """the transform kernel's groups of 8 batches (one CSPRNG block per draw serves a group): 1 batch, 7, 8, 9, 17 - ragged
    groups, zero padding of the last batch (batched.rs:37-43), several participants with distinct streams, large stream ids"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    k, t, n = 40, 23, 242
    w2, w3 = _root(P62, k + t + 1), _root(P62, n + 1)
    gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, P62, w2, w3))
    gen.set_drbg_key(KEY)
    rng = np.random.default_rng(dim)
    P, first = 3, (1 << 40) + 12345
    B = gen.batch_count(dim)
    stride = dim + 3                                               # odd participant stride, unaligned rows
    sec = rng.integers(-(1 << 62), 1 << 62, size=(P, stride), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    Bs = B + 5
    d_out = DeviceBuffer(P * n * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, stride, d_out.ptr, n * Bs, Bs, first_participant=first)
    out = d_out.to_numpy().reshape(P, n, Bs)
    for q in range(P):
        want = coracle.packed_generate_csprng(P62, k, t, n, w2, w3, sec[q, :dim], coracle.drbg_fill(KEY, first + q, B, t, P62),
                                              gen.csprng_share_map())
        assert np.array_equal(out[q, :, :B], want), f"participant {q}"
    assert not out[:, :, B:].any()


@pytest.mark.parametrize("above", [False, True])
def test_transform_path_modulus_bound(gpu, above):
    """round 2's transform kernel needed p < 2^62 - 2^31 (signed limbs); round 3's unsigned form takes every modulus the
    library admits (p < 2^62: 4p < 2^64).  Primes on either side of the old bound, special and random operands - bit-exact"""
    from sda_amd import crypto
    from oracle import coracle
    k, t, n = 40, 23, 242
    step = 64 * 243
    bound = (1 << 62) - (1 << 31)
    p = (bound // step) * step + 1 + (step if above else 0)            # the nearest p = 1 mod step on either side of the bound
    for _ in range(100000):                                             # (the window above the bound holds 138k candidates)
        if _is_prime(p):
            break
        p += step if above else -step
    assert _is_prime(p) and (p >= bound) == above and p < (1 << 62) and (p - 1) % step == 0
    g = next(g for g in range(2, 500) if pow(g, (p - 1) // 2, p) != 1 and pow(g, (p - 1) // 3, p) != 1)
    w2, w3 = pow(g, (p - 1) // 64, p), pow(g, (p - 1) // 243, p)
    assert pow(w2, 32, p) != 1 and pow(w3, 81, p) != 1
    rng = np.random.default_rng(5)
    dim = 40 * 11 + 7
    gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, p, w2, w3))
    B = gen.batch_count(dim)
    special = np.array([0, 1, p - 1, (p - 1) // 2, (p + 1) // 2, -p, p, -(1 << 62), (1 << 62) - 1], dtype=np.int64)
    secrets = rng.choice(special, size=dim)
    rand = rng.choice(special, size=B * t)
    assert np.array_equal(gen.generate(secrets, rand), coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand))
    secrets = rng.integers(0, p, size=dim, dtype=np.int64)
    rand = rng.integers(0, p, size=B * t, dtype=np.int64)
    assert np.array_equal(gen.generate(secrets, rand), coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand))


def test_small_prime_packed_matches_tss_fft_path(gpu):
    """p = 433 (full_loop.rs:57-64): the matrix form on the GPU equals the recalled tss FFT path."""
    from sda_amd import crypto
    from oracle import pyoracle as po
    pss = po.PackedSecretSharing(4, 8, 3, 433, 354, 150)
    rng = np.random.default_rng(1)
    secrets = rng.integers(0, 433, size=30).tolist()
    rand = rng.integers(0, 432, size=40).tolist()
    got = crypto.ShareGenerator(crypto.PackedShamir(3, 8, 4, 433, 354, 150)).generate(secrets, rand)
    for b in range(10):
        fft = pss.share_fft(secrets[3 * b:3 * b + 3], rand[4 * b:4 * b + 4], "rust_signed")
        assert [int(got[j][b]) for j in range(8)] == [v % 433 for v in fft]


# ---- device CSPRNG (sda-drbg-v1) ----------------------------------------------------------------------
@pytest.mark.parametrize("case", range(12))
def test_drbg_matches_spec(gpu, case):
    """Additive shares 0..n-2 ARE the raw draws, so generate(rand=NULL) exposes the CSPRNG stream."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    g = load_golden("drbg.json")
    c = g["cases"][case]
    T, B, m = c["T"], c["batches"], c["modulus"]
    gen = crypto.ShareGenerator(crypto.Additive(T + 1, m))
    gen.set_drbg_key(bytes.fromhex(g["key_hex"]))
    gen.set_drbg_rounds(c["rounds"])                      # accepted in deterministic mode only
    secrets = DeviceBuffer.from_numpy(np.zeros(B + (B & 1), dtype=np.int64))
    Bs = B + (B & 1)
    out = DeviceBuffer((T + 1) * Bs)
    gen.generate_batch_dev(secrets.ptr, 1, B, Bs, out.ptr, (T + 1) * Bs, Bs, first_participant=c["stream"])
    got = out.to_numpy().reshape(T + 1, Bs)[:, :B]
    want = np.array(c["values"], dtype=np.int64).reshape(B, T).T
    assert np.array_equal(got[:T], want)
    assert np.array_equal(got[T], np.mod(-want.astype(object).sum(axis=0), m).astype(np.int64))


@pytest.mark.parametrize("case", range(5))
def test_csprng_share_map_golden(gpu, case):
    """a generator without injected randomness against the committed share-map cases (tests/golden/drbg.json): the draws
    are shares 0..t-1, the other rows the oracle's interpolation; tss's map with the implied randomness gives the same
    shares; both maps through host and device entry points."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    g = load_golden("drbg.json")
    c = g["share_map_cases"][case]
    k, t, n, m = c["secret_count"], c["privacy_threshold"], c["share_count"], c["modulus"]
    gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, m, c["omega_secrets"], c["omega_shares"]))
    gen.set_drbg_key(bytes.fromhex(g["key_hex"]))
    assert gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC
    secrets = np.array(c["secrets"], dtype=np.int64)
    want = np.array(c["systematic_shares"], dtype=np.int64)
    B = want.shape[1]
    d_sec = DeviceBuffer.from_numpy(secrets)
    Bs = B + (B & 1)
    d_out = DeviceBuffer(n * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, 1, len(secrets), len(secrets), d_out.ptr, n * Bs, Bs, first_participant=c["stream"])
    assert np.array_equal(d_out.to_numpy().reshape(n, Bs)[:, :B], want)
    assert np.array_equal(gen.generate(secrets, np.array(c["implied_tss_randomness"], dtype=np.int64)), want)
    gen.set_csprng_share_map(gen.SHARE_MAP_TSS_NODES)
    d_out.zero()
    gen.generate_batch_dev(d_sec.ptr, 1, len(secrets), len(secrets), d_out.ptr, n * Bs, Bs, first_participant=c["stream"])
    assert np.array_equal(d_out.to_numpy().reshape(n, Bs)[:, :B],
                          coracle_generate(m, k, t, n, c["omega_secrets"], c["omega_shares"], secrets, c["draws"]))


def coracle_generate(p, k, t, n, w2, w3, secrets, rand):
    from oracle import coracle
    return coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)


def test_drbg_rejection_path(gpu):
    """A modulus with a large Lemire rejection zone forces the retry stream; C oracle has the same spec."""
    from sda_amd import crypto
    from oracle import coracle
    m = (1 << 61) + 1                       # 2^64 mod m is ~2^61-ish: heavy rejection
    gen = crypto.ShareGenerator(crypto.Additive(3, m))
    gen.set_drbg_key(KEY)
    dim = 4001
    got = gen.generate(np.zeros(dim, dtype=np.int64))        # stream 0
    want = coracle.drbg_fill(KEY, 0, dim, 2, m).reshape(dim, 2).T
    assert np.array_equal(got[:2], want)
    got2 = gen.generate(np.zeros(dim, dtype=np.int64))       # stream 1: fresh randomness per call
    assert np.array_equal(got2[:2], coracle.drbg_fill(KEY, 1, dim, 2, m).reshape(dim, 2).T)
    assert not np.array_equal(got, got2)


@pytest.mark.parametrize("m,T", [(8355709, 2), (0x7F7F7F, 3), (5038849, 5)])
def test_drbg_paired_rule_at_volume(gpu, m, T):
    """the PAIRED rule (moduli <= 0x7F7F7F, one candidate word -> two draws): millions of pairs, so the rare rejections
    (below 2^-18 per pair; tests/test_oracle.py locates them) and the retry stream happen on the device too; odd T = a last
    pair whose second element is discarded"""
    from sda_amd import crypto
    from oracle import coracle
    gen = crypto.ShareGenerator(crypto.Additive(T + 1, m))
    gen.set_drbg_key(KEY)
    dim = (6 << 20) // ((T + 1) // 2)
    got = gen.generate(np.zeros(dim, dtype=np.int64))        # stream 0
    want = coracle.drbg_fill(KEY, 0, dim, T, m).reshape(dim, T).T
    assert np.array_equal(got[:T], want)


def test_call_key_derivation_and_stream_hygiene(gpu):
    """The CSPRNG uniqueness contract of sda_hip.h: every drawing call of a handle runs under its own key
    KDF(master, call index), so repeating `first_participant` never repeats a keystream; deterministic streams exist
    only after set_drbg_key; stream ids are 56 bits; the round count cannot be lowered on a production handle."""
    from sda_amd import crypto, capi
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    m, T, B, P = P62, 2, 1000, 3
    sch = crypto.Additive(T + 1, m)
    secrets = DeviceBuffer.from_numpy(np.zeros(P * B, dtype=np.int64))
    out = DeviceBuffer(P * (T + 1) * B)

    def draws(gen, first):
        gen.generate_batch_dev(secrets.ptr, P, B, B, out.ptr, (T + 1) * B, B, first_participant=first)
        return out.to_numpy().reshape(P, T + 1, B)[:, :T, :].copy()

    def want(key, first):
        return np.stack([coracle.drbg_fill(key, first + p, B, T, m).reshape(B, T).T for p in range(P)])

    master = bytes((7 * i + 3) & 0xFF for i in range(32))
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_master_key(master)
    a, b, c = draws(gen, 5), draws(gen, 5), draws(gen, 0)          # the same first_participant twice, then another
    assert np.array_equal(a, want(coracle.drbg_call_key(master, 0), 5))
    assert np.array_equal(b, want(coracle.drbg_call_key(master, 1), 5))
    assert np.array_equal(c, want(coracle.drbg_call_key(master, 2), 0))
    assert not np.array_equal(a, b)
    # the host form draws a call key too (call 3), stream 0
    h = gen.generate(np.zeros(B, dtype=np.int64))
    assert np.array_equal(h[:T], want(coracle.drbg_call_key(master, 3), 0)[0])
    # production handle (OS-entropy master key): identical arguments, different randomness
    prod = crypto.ShareGenerator(sch)
    assert not np.array_equal(draws(prod, 0), draws(prod, 0))
    with pytest.raises(capi.SdaError) as e:
        prod.set_drbg_rounds(8)
    assert e.value.code == capi.ERR_STATE
    # deterministic mode: the caller's ids select the streams (and DO repeat)
    det = crypto.ShareGenerator(sch)
    det.set_drbg_key(master)
    assert np.array_equal(draws(det, 5), draws(det, 5))
    assert np.array_equal(draws(det, 5), want(master, 5))
    # 56-bit stream ids
    for first in (1 << 56, (1 << 56) - 1, (1 << 64) - 1):
        with pytest.raises(capi.SdaError) as e:
            draws(det, first)
        assert e.value.code == capi.ERR_INVALID_ARGUMENT
    draws(det, (1 << 56) - P)                                        # the last admissible ids
    # the masker obeys the same contract
    mk = crypto.SecretMasker(crypto.Full(m))
    mk.set_drbg_master_key(master)
    m1, _ = mk.mask(np.zeros(B, dtype=np.int64))
    m2, _ = mk.mask(np.zeros(B, dtype=np.int64))
    assert np.array_equal(m1, coracle.drbg_fill(coracle.drbg_call_key(master, 0), 0, B, 1, m))
    assert np.array_equal(m2, coracle.drbg_fill(coracle.drbg_call_key(master, 1), 0, B, 1, m))
    with pytest.raises(ValueError):
        mk.set_drbg_key(b"short")


def test_packed_generate_drbg_vs_oracle(gpu):
    from sda_amd import crypto
    from oracle import coracle
    for (k, t, n, o2, o3) in [s for s in PACKED_SHAPES if s[1] > 0]:           # every shape that draws randomness
        sch = crypto.PackedShamir(k, n, t, P62, W[o2], W[o3])
        gen = crypto.ShareGenerator(sch)
        gen.set_drbg_key(KEY)
        dim = 3001
        secrets = coracle.fill_synthetic(1, dim, 0, 1, P62)[0]
        B = gen.batch_count(dim)
        rnd = coracle.drbg_fill(KEY, 0, B, t, P62)
        # the default map of every matrix-form kernel: the draws ARE shares 0..t-1 (include/sda_hip.h "CSPRNG share map") ...
        assert gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC
        got = gen.generate(secrets)
        assert np.array_equal(got[:t], rnd.reshape(B, t).T), (k, t)
        sysm, implied = coracle.packed_generate_systematic(P62, k, t, n, W[o2], W[o3], secrets, rnd, want_implied=True)
        assert np.array_equal(got, sysm), (k, t)
        # ... which is a tss sharing of the same secrets: tss's own map reproduces it from the implied randomness
        assert np.array_equal(gen.generate(secrets, implied), got), (k, t)
        # and tss's map on request (the round-3 behaviour; what the transform kernel always does)
        gen.set_csprng_share_map(gen.SHARE_MAP_TSS_NODES)
        rnd1 = coracle.drbg_fill(KEY, 1, B, t, P62)                     # the second host call that draws: stream id 1
        assert np.array_equal(gen.generate(secrets), coracle.packed_generate(P62, k, t, n, W[o2], W[o3], secrets, rnd1)), (k, t)


# ---- combiner -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,L,q", [(2, 4, 433), (1, 1, P62), (7, 1001, P62), (300, 64, P62), (5000, 6, P62),
                                   (33, 4097, (1 << 62) - 57), (3, 2, 2)])
def test_combine_vs_oracle(gpu, P, L, q):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(P * 7 + L)
    rows = rng.integers(-(q - 1), q, size=(P, L), dtype=np.int64)          # the reference's (-q, q) domain
    comb = crypto.ShareCombiner(crypto.Additive(3, q))
    got = comb.combine(list(rows))
    assert np.array_equal(got, coracle.combine(q, rows, mode=0))
    assert np.array_equal(got, np.mod(coracle.combine(q, rows, mode=1), q))   # == reference mod q
    wild = rng.integers(-(1 << 63), (1 << 63) - 1, size=(P, L), dtype=np.int64)   # any i64 is accepted
    assert np.array_equal(comb.combine(list(wild)), coracle.combine(q, wild, mode=0))


def test_combine_edge_cases(gpu):
    from sda_amd import capi, crypto
    comb = crypto.ShareCombiner(crypto.PackedShamir(3, 8, 1, P62, W[8], W[9]))
    assert comb.combine([]).size == 0                                       # combiner.rs:17
    assert comb.combine([[], []]).size == 0
    with pytest.raises(capi.SdaError) as e:
        comb.combine([[1, 2, 3], [1, 2]])
    assert e.value.code == capi.ERR_WRONG_DIMENSION and "Wrong dimension" in e.value.message   # combiner.rs:21
    assert list(comb.combine([[5, -1, P62]])) == [5, P62 - 1, 0]


def test_streaming_combiner_equals_one_shot(gpu):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(3)
    P, L = 257, 3001
    rows = rng.integers(0, P62, size=(P, L), dtype=np.int64)
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    comb.begin(L)
    for lo in range(0, P, 50):
        comb.update(rows[lo:lo + 50])
    assert np.array_equal(comb.finish(L), coracle.combine(P62, rows))


def test_combiner_dev_jobs_layout(gpu):
    """[jobs][rows][L] resident in HBM, the layout the snapshot transposition yields (stores.rs:86-101)."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(4)
    jobs, P, L = 8, 40, 1234
    data = rng.integers(0, P62, size=(jobs, P, L), dtype=np.int64)
    d = DeviceBuffer.from_numpy(data)
    out = DeviceBuffer(jobs * L)
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    comb.begin_dev(jobs, L)
    comb.update_dev(d.ptr, P * L, P // 2, L)
    comb.update_dev(d.at((P // 2) * L), P * L, P - P // 2, L)
    comb.finish_dev(out.ptr)
    got = out.to_numpy().reshape(jobs, L)
    for j in range(jobs):
        assert np.array_equal(got[j], coracle.combine(P62, data[j]))


# ---- reconstruct ----------------------------------------------------------------------------------------
def test_packed_reconstruct_vs_oracle_and_errors(gpu):
    from sda_amd import capi, crypto
    from oracle import coracle
    k, t, n = 3, 4, 8
    sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
    dim = 1000
    rng = np.random.default_rng(11)
    secrets = rng.integers(0, P62, size=dim, dtype=np.int64)
    B = (dim + k - 1) // k
    rand = rng.integers(0, P62, size=B * t, dtype=np.int64)
    shares = coracle.packed_generate(P62, k, t, n, W[8], W[9], secrets, rand)
    rec = crypto.SecretReconstructor(sch, dim)
    for subset in ([0, 1, 2, 3, 4, 5, 6, 7], [7, 5, 3, 1, 0, 2, 6], [1, 2, 3, 4, 5, 6, 7]):
        got = rec.reconstruct([(c, shares[c]) for c in subset])
        assert np.array_equal(got, secrets)
        assert np.array_equal(got, coracle.packed_reconstruct(P62, k, t, W[8], W[9], dim, subset, shares[subset]))
    with pytest.raises(capi.SdaError) as e:
        rec.reconstruct([(c, shares[c]) for c in range(6)])
    assert e.value.code == capi.ERR_NOT_ENOUGH_SHARES and "Not enough shares to reconstruct" in e.value.message
    with pytest.raises(capi.SdaError) as e:                                  # duplicate clerk index
        rec.reconstruct([(c, shares[c]) for c in [0, 1, 2, 3, 4, 5, 5]])
    assert e.value.code == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.SdaError) as e:                                  # a short row: index OOB in batched.rs:84
        rec.reconstruct([(c, shares[c][:-1]) for c in range(7)])
    assert e.value.code == capi.ERR_ASSERTION


def test_additive_reconstruct(gpu):
    from sda_amd import capi, crypto
    rec = crypto.SecretReconstructor(crypto.Additive(3, 433), 99)            # configured dimension is ignored
    assert list(rec.reconstruct([(0, [400, 1]), (1, [30, 2]), (2, [4, -3])])) == [1, 0]
    assert rec.reconstruct([]).size == 0                                     # additive.rs:57-60
    with pytest.raises(capi.SdaError) as e:
        rec.reconstruct([(0, [1, 2]), (1, [1])])
    assert e.value.code == capi.ERR_MISMATCHING_DIMENSION and "Mismatching dimension" in e.value.message


# ---- masking ----------------------------------------------------------------------------------------------
def test_chacha_masks_kats(gpu):
    from sda_amd import crypto
    k = load_golden("kats.json")["kats"]
    for name in ("C2_masks_seed0", "C3_masks_seed1234"):
        c = k[name]
        comb = crypto.MaskCombiner(crypto.ChaCha(c["modulus"], 8, 128))
        assert list(map(int, comb.combine([c["seed"]]))) == c["expected"]


@pytest.mark.parametrize("q,dim,seeds", [(433, 1000, 5), (P62, 4099, 9), (P62, 8, 1), (97, 3, 300),
                                         ((1 << 61) + 1, 3000, 6),            # ~12% rejection: exact-order path for all
                                         ((1 << 62) - (1 << 49), 2000, 60),   # 2^-13 rejection: fast path + fix-ups
                                         ((1 << 62) - (1 << 51), 1500, 300),  # ~0.7 rejections per seed: shift pass (<= 3) and exact-order kernel (> 3)
                                         ((1 << 62) - (1 << 51), 40, 3000),   # rejections near the end of short streams: the tail walk
                                         ((1 << 61) + 1, 8, 500)])            # 12% rejection on 8 candidates: tails that meet further rejections
def test_chacha_combine_vs_oracle(gpu, q, dim, seeds):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(dim + seeds)
    S = rng.integers(0, 1 << 32, size=(seeds, 4), dtype=np.int64)
    got = crypto.MaskCombiner(crypto.ChaCha(q, dim, 128)).combine(list(S))
    assert np.array_equal(got, coracle.chacha_combine(S, q, dim))


def test_chacha_mask_roundtrip_and_seed_forms(gpu):
    from sda_amd import crypto
    from oracle import coracle
    q, dim = P62, 1237
    sch = crypto.ChaCha(q, dim, 128)
    secrets = np.arange(dim, dtype=np.int64) * 3 - 50
    m = crypto.SecretMasker(sch)
    seed, masked = m.mask(secrets, [1, 2, 3, 4])
    assert list(seed) == [1, 2, 3, 4]                                        # chacha.rs:48-50
    mask = coracle.chacha_expand([1, 2, 3, 4], q, dim)
    assert np.array_equal(masked, coracle.addsub(secrets, mask, q))
    seed2, masked2 = m.mask(secrets)                                         # OS-entropy seed
    assert seed2.size == 4 and all(0 <= int(w) < (1 << 32) for w in seed2)
    mask2 = crypto.MaskCombiner(sch).combine([seed2])
    assert np.array_equal(crypto.SecretUnmasker(sch).unmask((mask2, masked2)), np.mod(secrets, q))
    # 256-bit and 64-bit seeds; words beyond 8 are ignored by rand 0.3
    for words in ([9, 8, 7, 6, 5, 4, 3, 2], [1, 2], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]):
        got = crypto.MaskCombiner(crypto.ChaCha(q, 64, 32 * len(words))).combine([words])
        assert np.array_equal(got, coracle.chacha_expand(words, q, 64))
    with pytest.raises(AssertionError):                                      # assert_eq!, chacha.rs:26
        m.mask(secrets[:-1], [1, 2, 3, 4])
    assert list(crypto.MaskCombiner(crypto.ChaCha(q, 5, 128)).combine([])) == [0] * 5   # chacha.rs:58


def test_full_and_none_masking(gpu):
    from sda_amd import crypto
    from oracle import coracle
    q, dim = P62, 1001
    rng = np.random.default_rng(8)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    draws = rng.integers(0, q, size=dim, dtype=np.int64)
    full = crypto.Full(q)
    mask, masked = crypto.SecretMasker(full).mask(secrets, draws)
    assert np.array_equal(mask, draws) and np.array_equal(masked, coracle.addsub(secrets, draws, q))
    m = crypto.SecretMasker(full)
    m.set_drbg_key(KEY)
    mask2, masked2 = m.mask(secrets)                                         # device CSPRNG
    assert np.array_equal(mask2, coracle.drbg_fill(KEY, 0, dim, 1, q))
    assert np.array_equal(crypto.SecretUnmasker(full).unmask((mask2, masked2)), np.mod(secrets, q))
    masks = rng.integers(0, q, size=(12, dim), dtype=np.int64)
    assert np.array_equal(crypto.MaskCombiner(full).combine(list(masks)), coracle.combine(q, masks))
    with pytest.raises(AssertionError):                                      # full.rs:43
        crypto.MaskCombiner(full).combine([[1, 2], [1]])
    with pytest.raises(AssertionError):                                      # full.rs:58
        crypto.SecretUnmasker(full).unmask(([1], [1, 2]))
    none = crypto.NoMask()
    mk, ms = crypto.SecretMasker(none).mask([5, -7, 9])
    assert mk.size == 0 and list(ms) == [5, -7, 9]                            # none.rs:13-19: untouched
    assert crypto.MaskCombiner(none).combine([[], []]).size == 0
    with pytest.raises(AssertionError):                                      # none.rs:23
        crypto.MaskCombiner(none).combine([[1]])
    with pytest.raises(AssertionError):                                      # none.rs:30
        crypto.SecretUnmasker(none).unmask(([1], [1]))
    assert list(crypto.SecretUnmasker(none).unmask(([], [3, -4]))) == [3, -4]


@pytest.mark.parametrize("P,dim,pad", [(1, 1, 0), (5, 1001, 3), (40, 4096, 0), (7, 777, 1)])
def test_masked_participation_on_device(gpu, P, dim, pad):
    """participate.rs:52-76 for a device-resident tile: full masking (masks from the device CSPRNG, stream = participant)
    then additive share generation of the MASKED secrets; recipient side receive.rs:113-152: combine masks, combine
    clerk sums, reconstruct, unmask == sum of the secrets.  Every stage against the oracle."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    q = P62
    rng = np.random.default_rng(P * 31 + dim)
    stride = dim + pad
    sec = np.zeros((P, stride), dtype=np.int64)
    sec[:, :dim] = rng.integers(-(1 << 62), 1 << 62, size=(P, dim), dtype=np.int64)       # any i64 is accepted
    d_sec = DeviceBuffer.from_numpy(sec)
    masker = crypto.SecretMasker(crypto.Full(q))
    masker.set_drbg_key(KEY)
    d_mask, d_masked = DeviceBuffer(P * stride).zero(), DeviceBuffer(P * stride).zero()
    masker.mask_batch_dev(d_sec.ptr, P, dim, stride, d_mask.ptr, stride, d_masked.ptr, stride, first_participant=100)
    masks = d_mask.to_numpy().reshape(P, stride)
    masked = d_masked.to_numpy().reshape(P, stride)
    for p in (0, P // 2, P - 1):
        want = coracle.drbg_fill(KEY, 100 + p, dim, 1, q)
        assert np.array_equal(masks[p, :dim], want)
        assert np.array_equal(masked[p, :dim], coracle.addsub(sec[p, :dim], want, q))
    assert not masks[:, dim:].any() and not masked[:, dim:].any()
    # shares of the masked secrets, clerk sums, reconstruction
    sch = crypto.Additive(3, q)
    gen = crypto.ShareGenerator(sch); gen.set_drbg_key(KEY)
    Bs = dim + (dim & 1)
    shares = DeviceBuffer(3 * P * Bs).zero()
    gen.generate_batch_dev(d_masked.ptr, P, dim, stride, shares.ptr, Bs, P * Bs, first_participant=100)
    comb = crypto.ShareCombiner(sch)
    comb.begin_dev(3, dim)
    comb.update_dev(shares.ptr, P * Bs, P, Bs)
    sums = DeviceBuffer(3 * dim)
    comb.finish_dev(sums.ptr)
    rec = crypto.SecretReconstructor(sch, dim)
    masked_total = DeviceBuffer(dim)
    rec.reconstruct_dev([0, 1, 2], sums.ptr, dim, dim, masked_total.ptr, dim)
    assert np.array_equal(masked_total.to_numpy(), coracle.combine(q, masked[:, :dim]))
    # recipient: the P mask vectors are combined like shares (full.rs:37-52), then unmask
    mc = crypto.ShareCombiner(sch)
    mc.begin_dev(1, dim)
    mc.update_dev(d_mask.ptr, 0, P, stride)
    mask_total = DeviceBuffer(dim)
    mc.finish_dev(mask_total.ptr)
    assert np.array_equal(mask_total.to_numpy(), crypto.MaskCombiner(crypto.Full(q)).combine(list(masks[:, :dim])))
    out = DeviceBuffer(dim)
    crypto.SecretUnmasker(crypto.Full(q)).unmask_dev(mask_total.ptr, masked_total.ptr, dim, out.ptr)
    assert np.array_equal(out.to_numpy(), coracle.combine(q, sec[:, :dim]))
    # None: identity; ChaCha: refused with a pointer to the host form
    none = crypto.SecretMasker(crypto.NoMask())
    d2 = DeviceBuffer(P * stride).zero()
    none.mask_batch_dev(d_sec.ptr, P, dim, stride, 0, 0, d2.ptr, stride)
    assert np.array_equal(d2.to_numpy().reshape(P, stride)[:, :dim], sec[:, :dim])


@pytest.mark.parametrize("q,dim,P,bits", [(P62, 1237, 9, 128), (433, 40, 50, 128), (P62, 8, 3, 64),
                                          ((1 << 62) - (1 << 51), 1500, 120, 128),   # ~0.7 rejections per seed: shift + exact
                                          ((1 << 62) - (1 << 51), 40, 800, 256),     # rejections near the end: tail walk
                                          ((1 << 61) + 1, 8, 300, 128),              # 12 % rejection on 8 candidates
                                          ((1 << 61) + 1, 3000, 4, 128)])            # rejections are the rule: exact order for all
def test_chacha_masking_of_a_device_tile(gpu, q, dim, P, bits):
    """chacha.rs:24-54 for a device-resident tile: every participant gets an OS-entropy seed (returned as its mask,
    chacha.rs:48-50) and masked[p][i] = (secret + i-th rand-0.3 gen_range value of that seed) mod q, bit-exact against
    the oracle's expansion of the seeds read back - through the fast pass, the parallel shift repair, the exact-order
    walk and the all-exact form; the recipient's MaskCombiner over the same seeds then unmasks the sum."""
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(dim * 7 + P)
    nw = (bits + 31) // 32
    stride, mstride = dim + 3, nw + 1
    sec = np.zeros((P, stride), dtype=np.int64)
    sec[:, :dim] = rng.integers(-(1 << 62), 1 << 62, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    d_seed = DeviceBuffer(P * mstride).zero()
    d_masked = DeviceBuffer(P * stride).zero()
    sch = crypto.ChaCha(q, dim, bits)
    crypto.SecretMasker(sch).mask_batch_dev(d_sec.ptr, P, dim, stride, d_seed.ptr, mstride, d_masked.ptr, stride)
    seeds = d_seed.to_numpy().reshape(P, mstride)
    masked = d_masked.to_numpy().reshape(P, stride)
    assert not seeds[:, nw:].any() and not masked[:, dim:].any()
    assert ((seeds[:, :nw] >= 0) & (seeds[:, :nw] < (1 << 32))).all()          # u32 words (chacha.rs:30-33)
    assert len({tuple(r) for r in seeds[:, :nw]}) == P                           # fresh entropy per participant
    for p in range(P):
        mask = coracle.chacha_expand(seeds[p, :nw], q, dim)
        assert np.array_equal(masked[p, :dim], coracle.addsub(sec[p, :dim], mask, q)), p
    # recipient: combine the seeds, combine the masked vectors, unmask == sum of the secrets
    total_mask = crypto.MaskCombiner(sch).combine(np.ascontiguousarray(seeds[:, :nw]))
    total_masked = coracle.combine(q, masked[:, :dim])
    assert np.array_equal(crypto.SecretUnmasker(sch).unmask((total_mask, total_masked)), coracle.combine(q, sec[:, :dim]))
    with pytest.raises(AssertionError):                                          # assert_eq!, chacha.rs:26
        try:
            crypto.SecretMasker(sch).mask_batch_dev(d_sec.ptr, P, dim - 1 if dim > 1 else 2, stride, d_seed.ptr, mstride,
                                                    d_masked.ptr, stride)
        except capi.SdaError as e:
            if e.code == capi.ERR_ASSERTION:
                raise AssertionError(e.message)
            raise


def test_scheme_validation(gpu):
    from sda_amd import capi, crypto
    for bad in (crypto.Additive(3, 1), crypto.Additive(0, 433), crypto.Additive(3, 1 << 62),
                crypto.PackedShamir(3, 8, 4, 435, 354, 150),           # composite modulus
                crypto.PackedShamir(3, 8, 4, 433, 1, 150),             # omega_secrets of order 1: nodes collide
                crypto.PackedShamir(0, 8, 4, 433, 354, 150)):
        with pytest.raises(capi.SdaError):
            crypto.ShareGenerator(bad)
    # the descriptor is network-supplied: u64 wrap-arounds must not pass, for any role
    for bad in (crypto.PackedShamir((1 << 64) - 1, 8, 2, 433, 354, 150),   # k + t wraps to 1
                crypto.PackedShamir(3, 8, (1 << 64) - 2, 433, 354, 150),   # k + t wraps to 1
                crypto.PackedShamir(3, 1 << 32, 4, 433, 354, 150),         # share_count would truncate to 0
                crypto.PackedShamir(5, 8, 4, 433, 354, 150)):              # share_count < t + k: never reconstructible
        with pytest.raises(capi.SdaError):
            crypto.ShareGenerator(bad)
        with pytest.raises(capi.SdaError):
            crypto.SecretReconstructor(bad, 10)
    with pytest.raises(capi.SdaError):
        crypto.SecretReconstructor(crypto.Additive(0, 433), 10)
    with pytest.raises(capi.SdaError):
        crypto.SecretReconstructor(crypto.Additive(1 << 40, 433), 10)
    crypto.ShareGenerator(crypto.PackedShamir(3, 8, 0, 433, 354, 150))     # t = 0 is accepted (documented)
    # host finish on a multi-job combiner is refused instead of overrunning its buffer
    comb = crypto.ShareCombiner(crypto.Additive(3, 433))
    comb.begin_dev(2, 8)
    out = np.zeros(16, dtype=np.int64)
    assert capi.load().sda_share_combiner_finish(comb._h, out.ctypes.data_as(capi.c_i64p)) == capi.ERR_STATE
    comb.begin(8)
    with pytest.raises(ValueError):
        comb.finish(4)
    assert comb.finish().shape == (8,)


# ---- BASELINE-size property tests (size-independent invariants; data stays in HBM) ------------------------
@pytest.mark.parametrize("shape", ["additive_n3", "packed_k3_t1_n8", "packed_k8_t2_n26", "narrow_k3_t4_n8_p31", "narrow_k8_t7_n26_p31",
                                   "narrow_k3_t4_n8_p433"])
def test_full_dimension_roundtrip(gpu, shape):
    """dim = 1,048,576 (BASELINE configs 2-4), P participants on the device CSPRNG:
    reconstruct(combine(generate(x_p))) == sum_p x_p mod q, for a strict subset of clerks where the
    scheme allows it; plus bit-exact share spot-checks against the oracle on sampled batches."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    dim, P = 1 << 20, 24
    q = P62
    if shape.startswith("narrow"):
        # the reference's own domain (tss multiplies i64 residues without widening): tss-valid shapes over the largest prime
        # = 1 mod 432 below 2^31 and over full_loop.rs's p = 433, through the one-limb kernels (generate, sums, narrow reveal)
        k, t, n = (3, 4, 8) if "k3" in shape else (8, 7, 26)
        q = 433 if shape.endswith("p433") else 2147482801
        w2 = 354 if q == 433 else _root(q, k + t + 1)
        w3 = 150 if q == 433 else _root(q, n + 1)
        sch = crypto.PackedShamir(k, n, t, q, w2, w3)
        subset = sorted(np.random.default_rng(5).choice(n, size=k + t, replace=False).tolist())
    elif shape == "additive_n3":
        sch, k, t, n = crypto.Additive(3, P62), 1, 2, 3
        subset = [0, 1, 2]
    elif shape == "packed_k3_t1_n8":
        k, t, n = 3, 1, 8
        sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
        subset = [6, 1, 4, 3]
    else:
        k, t, n = 8, 2, 26
        sch = crypto.PackedShamir(k, n, t, P62, W[16], W[27])
        subset = list(range(25, 15, -1))
    B = (dim + k - 1) // k
    Bs = B + (B & 1)
    lib = gpu
    secrets = DeviceBuffer(P * dim)
    from sda_amd.capi import check
    check(lib.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 100, 0x5DA5DA5DA5DA5DA5, q, None))
    shares = DeviceBuffer(n * P * Bs)                                        # job-major [n][P][Bs]
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    gen.generate_batch_dev(secrets.ptr, P, dim, dim, shares.ptr, Bs, P * Bs, first_participant=100)
    comb = crypto.ShareCombiner(sch)
    sums = DeviceBuffer(n * B)
    comb.begin_dev(n, B)
    comb.update_dev(shares.ptr, P * Bs, P, Bs)
    comb.finish_dev(sums.ptr)
    rec = crypto.SecretReconstructor(sch, dim)
    S = sums.to_numpy().reshape(n, B)
    got = rec.reconstruct([(c, S[c]) for c in subset])
    host_secrets = secrets.to_numpy().reshape(P, dim)
    assert np.array_equal(host_secrets[:2], coracle.fill_synthetic(2, dim, 100, 0x5DA5DA5DA5DA5DA5, q))
    assert np.array_equal(got, coracle.combine(q, host_secrets))             # == sum of secrets mod q
    # device-resident reconstruct from the first rows (contiguous [n'][B] in HBM)
    first = list(range(sch.reconstruction_threshold()))
    out = DeviceBuffer(dim)
    assert rec.reconstruct_dev(first, sums.ptr, B, B, out.ptr, dim) == dim
    assert np.array_equal(out.to_numpy(), got)
    # spot-check one participant's shares bit-exactly against the oracle
    p = 7
    rnd = coracle.drbg_fill(KEY, 100 + p, B, t, q)
    if shape == "additive_n3":
        want = coracle.additive_generate(q, n, host_secrets[p], rnd)
    else:
        want = coracle.packed_generate_csprng(q, k, t, n, sch.omega_secrets, sch.omega_shares, host_secrets[p], rnd, gen.csprng_share_map())
    all_shares = shares.to_numpy().reshape(n, P, Bs)
    assert np.array_equal(all_shares[:, p, :B], want)
    assert np.array_equal(S, np.stack([coracle.combine(q, all_shares[c, :, :B]) for c in range(n)]))


# ---- host mirror in C++ (the reference's host language is compiled) and multi-GPU helper ---------------------
def test_config5_dimension_16m_vs_oracle(gpu):
    """BASELINE config 5's dimension (16,777,216; k=3, t=1, n=8; B = 5,592,406 batches) through the HIP path against the
    oracle, bit for bit: every participant's shares (device CSPRNG streams reproduced by the oracle), the clerk sums,
    and the Lagrange reveal over the full 16 Mi secrets from a non-trivial clerk subset
    (batched.rs:18-53,68-97; combiner.rs:15-29; receive.rs:140-152)."""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    dim, k, t, n, P = 1 << 24, 3, 1, 8, 3
    sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
    B = (dim + k - 1) // k
    assert B == 5_592_406
    Bs = (B + 15) // 16 * 16
    d_sec = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(d_sec.ptr, P, dim, dim, 40, 77, P62, None))
    secrets = coracle.fill_synthetic(P, dim, 40, 77, P62)
    assert np.array_equal(d_sec.to_numpy().reshape(P, dim), secrets)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    d_sh = DeviceBuffer(n * P * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_sh.ptr, Bs, P * Bs, first_participant=1000)   # job-major [n][P][Bs]
    got = d_sh.to_numpy().reshape(n, P, Bs)
    assert gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC
    want = [coracle.packed_generate_systematic(P62, k, t, n, W[8], W[9], secrets[p], coracle.drbg_fill(KEY, 1000 + p, B, t, P62))
            for p in range(P)]
    for p in range(P):
        assert np.array_equal(got[:, p, :B], want[p]), f"participant {p}"
    assert not got[:, :, B:].any()                                       # the row padding is never written
    # clerk sums: separate launch and the dual-role launch (sum only), both vs the oracle's combine
    comb = crypto.ShareCombiner(sch)
    comb.begin_dev(n, B)
    comb.update_dev(d_sh.ptr, P * Bs, P, Bs)
    d_sums = DeviceBuffer(n * B)
    comb.finish_dev(d_sums.ptr)
    sums = d_sums.to_numpy().reshape(n, B)
    for c in range(n):
        assert np.array_equal(sums[c], coracle.combine(P62, np.stack([want[p][c] for p in range(P)]))), f"clerk {c}"
    comb.begin_dev(n, B)
    gen.generate_combine_dev(comb, 0, 0, dim, dim, 0, Bs, P * Bs, d_prev=d_sh.ptr, prev_participants=P)
    d_sums2 = DeviceBuffer(n * B)
    comb.finish_dev(d_sums2.ptr)
    assert np.array_equal(d_sums2.to_numpy().reshape(n, B), sums)
    # reveal over the full dimension from clerks {6, 1, 4, 3}
    subset = [6, 1, 4, 3]
    rows = DeviceBuffer.from_numpy(np.ascontiguousarray(sums[subset]))
    d_out = DeviceBuffer(dim)
    rec = crypto.SecretReconstructor(sch, dim)
    assert rec.reconstruct_dev(subset, rows.ptr, B, B, d_out.ptr, dim) == dim
    out = d_out.to_numpy()
    assert np.array_equal(out, coracle.packed_reconstruct(P62, k, t, W[8], W[9], dim, subset, sums[subset]))
    assert np.array_equal(out, coracle.combine(P62, secrets))           # == sum of the secrets mod p


def test_cpp_host_mirror_full_loop(gpu):
    """tests/cpp/full_loop.cpp: the reference's full_loop.rs scenarios through sda_amd/host/sda_crypto.hpp,
    with OS randomness like the reference's own tests."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "full_loop")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all scenarios OK" in out.stdout


def test_c_abi_walkthrough(gpu):
    """examples/c_abi_walkthrough.c: the README walkthrough and the packed-Shamir + ChaCha-mask loop through the raw
    C ABI from plain C99 (what a Rust / cgo binding would call)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "c_abi_walkthrough")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "c_abi_walkthrough: OK" in out.stdout, out.stdout + out.stderr


def test_modsum_parts_dev(gpu):
    """cross-GPU partial-sum reducer: 8 parts of (q-1) must not wrap (a plain u64 SUM would)."""
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(2)
    parts = rng.integers(0, P62, size=(8, 1001), dtype=np.int64)
    parts[:, 0] = P62 - 1
    d = DeviceBuffer.from_numpy(parts)
    out = DeviceBuffer(1001)
    check(gpu.sda_modsum_parts_dev(P62, d.ptr, 8, 1001, 1001, out.ptr, None))
    assert np.array_equal(out.to_numpy(), coracle.combine(P62, parts))
    assert out.to_numpy()[0] == (8 * (P62 - 1)) % P62


def test_comm_c_abi_single_rank_rccl(gpu, monkeypatch):
    """sda_comm_* / sda_modular_allreduce_dev through ctypes with one rank: the no-exchange shortcut, and - with
    SDA_FORCE_COLLECTIVES - the full RCCL path (grouped ncclSend/ncclRecv to itself, modular sum, gather) for ragged
    lengths.  Inputs are any i64; outputs canonical."""
    import ctypes as C
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    rng = np.random.default_rng(11)
    for force in (False, True):
        if force:
            set_knob("SDA_FORCE_COLLECTIVES", "1")      # (libsda_hip_test.so from here on: the communicator is made in the library that uses it)
        ident = (C.c_uint8 * 128)()
        check(gpu.sda_comm_unique_id(ident))
        comm = C.c_void_p()
        check(gpu.sda_comm_init(ident, 0, 1, C.byref(comm)))
        assert gpu.sda_comm_rank(comm) == 0 and gpu.sda_comm_world(comm) == 1 and gpu.sda_comm_rccl_version() > 20000
        try:
            for n in (1, 7, 1000, 22369 * 8 + 3):
                v = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
                d, o = DeviceBuffer.from_numpy(v), DeviceBuffer(n)
                check(gpu.sda_modular_allreduce_dev(comm, P62, d.ptr, n, o.ptr, None))
                check(gpu.sda_dev_synchronize())
                assert np.array_equal(o.to_numpy(), np.mod(v.astype(object), P62).astype(np.int64)), (force, n)
            check(gpu.sda_modular_allreduce_dev(comm, P62, None, 0, None, None))           # empty vector: nothing to do
            assert gpu.sda_modular_allreduce_dev(None, P62, None, 0, None, None) == capi_err()
            assert gpu.sda_comm_init(ident, 1, 1, C.byref(C.c_void_p())) == capi_err()    # rank out of range
        finally:
            gpu.sda_comm_free(comm)


def capi_err():
    from sda_amd import capi
    return capi.ERR_INVALID_ARGUMENT


def test_any_i64_canonicalisation_property(gpu):
    """the boundary accepts ANY i64 (SURVEY.md 8b 'Value domain'): outputs equal python's x % q."""
    from sda_amd import crypto
    edge = np.array([0, 1, -1, P62 - 1, P62, P62 + 1, -P62, 2 ** 63 - 1, -2 ** 63, -2 ** 63 + 1, 2 ** 62, -(2 ** 62)],
                    dtype=np.int64)
    for q in (433, P62, 2, (1 << 62) - 57):
        got = crypto.ShareCombiner(crypto.Additive(3, q)).combine([edge])
        assert got.tolist() == [int(x) % q for x in edge.tolist()]
        un = crypto.SecretUnmasker(crypto.Full(q)).unmask((edge, edge[::-1].copy()))
        assert un.tolist() == [(int(b) - int(a)) % q for a, b in zip(edge.tolist(), edge[::-1].tolist())]
    sch = crypto.PackedShamir(3, 8, 1, (1 << 62) - 57, pow(3, ((1 << 62) - 58) // 8, (1 << 62) - 57), 1)
    # omega_shares = 1 makes the share points collide with node 1: generation still defined, checked vs oracle
    from oracle import coracle
    p = (1 << 62) - 57
    w2 = sch.omega_secrets
    if len({pow(w2, e, p) for e in range(5)}) == 5:
        g = crypto.ShareGenerator(crypto.PackedShamir(3, 8, 1, p, w2, 3))
        rnd = np.array([5, -7], dtype=np.int64)
        got = g.generate(edge[:6], rnd)
        assert np.array_equal(got, coracle.packed_generate(p, 3, 1, 8, w2, 3, edge[:6], rnd))


def _run_bench(cmd, tmp_path, env=None, timeout=900, root=None):
    """run bench.py (any launcher form): stdout must be exactly ONE compact JSON line of at most 4096 bytes (the driver parses
    it; BENCH_r04 came back unparsed at 35 KB); the full record is read back from the --details file"""
    import json
    import subprocess
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    details = os.path.join(str(tmp_path), "bench_details.json")
    out = subprocess.run(cmd + ["--details", details], capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be the ONE JSON line (banners of gloo / RCCL belong on stderr)"
    assert len(lines[0].encode()) <= 4096, len(lines[0])
    line = json.loads(lines[0])
    full = json.load(open(details))
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["details"] == "bench_details.json"
    assert line["build_id"] == full["build_id"] and len(line["build_id"]) == 16
    return line, full, out


def test_bench_under_torchrun_single_rank_rccl(gpu, tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run), with one rank and the exchange
    forced through RCCL (the library's own communicator: ncclCommInitRank, grouped ncclSend/ncclRecv to itself, the
    modular-sum kernel, the gather).  The result must verify (reconstruct == sum of secrets)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDA_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--participants", "128", "--dim", "65536", "--no-cpu-baseline", "--no-additional"]
    line, d, _ = _run_bench(cmd, tmp_path, env=env, timeout=600)
    assert d["n_gpus"] == 1 and d["verified_reconstruct_equals_sum"] is True and line["verified_reconstruct_equals_sum"] is True
    assert d["roofline"]["bound"] in ("hbm", "valu") and d["value"] > 0
    # the kernel name is what the LIBRARY reports it launched (sda_debug_last_kernel), as rocprofv3 prints it
    assert line["roofline"]["kernel"] == "fused_packed_l31_kernel<3, 1, 20>" and line["config"]["library_path"] == "l31"
    # the machine-readable record of what carried the exchange: the library's communicator spans the one rank
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["unique_devices"] == 1 and d["rccl"]["path"].startswith("send/recv")
    assert d["rccl"]["comm_device"] == 0


def test_launch_slicing_over_the_grid_limit(gpu):
    """(participant, chunk) grids beyond HIP's 2^32 work-items per launch are issued in participant slices;
    results must not depend on the slicing (checked against the oracle on sampled participants)."""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    # additive, n = 1: share = secret mod q.  1024 workgroups per participant x 18000 participants =
    # 18.4 M workgroups > 2^24 (and twice that for the fill kernel), forcing several slices.  2 x 75 GB of HBM.
    P, dim = 18_000, 1 << 19
    gen = crypto.ShareGenerator(crypto.Additive(1, P62))
    secrets = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 5, 11, P62, None))      # sliced too
    out = DeviceBuffer(P * dim)
    gen.generate_batch_dev(secrets.ptr, P, dim, dim, out.ptr, dim, P * dim, first_participant=5)
    for p in (0, 8_191, 8_192, 16_383, 16_384, P - 1):
        want = coracle.fill_synthetic(1, dim, 5 + p, 11, P62)[0]
        assert np.array_equal(secrets.to_numpy(dim, p * dim), want)
        assert np.array_equal(out.to_numpy(dim, p * dim), want)


def test_mont64_kernel_equals_limb31_kernel(gpu, monkeypatch):
    """the superseded 64-bit Montgomery share-gen kernel (kept for A/B) stays bit-identical to the shipped one"""
    from sda_amd import crypto
    rng = np.random.default_rng(6)
    for (k, t, n, o2, o3) in [(3, 1, 8, 8, 9), (8, 2, 26, 16, 27)]:
        sch = crypto.PackedShamir(k, n, t, P62, W[o2], W[o3])
        secrets = rng.integers(-(1 << 63), (1 << 63) - 1, size=4099, dtype=np.int64)
        a = crypto.ShareGenerator(sch); a.set_drbg_key(KEY)
        set_knob("SDA_FORCE_MONT64", "1")
        b = crypto.ShareGenerator(sch); b.set_drbg_key(KEY)
        set_knob("SDA_FORCE_MONT64", 0)
        assert np.array_equal(a.generate(secrets), b.generate(secrets))


def _is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


@pytest.mark.parametrize("bits", [7, 13, 31, 32, 33, 47, 61, 62])
def test_packed_random_primes_and_roots(gpu, bits):
    """any odd prime < 2^62 and any omegas with distinct nodes: generate -> combine -> reconstruct vs the oracle"""
    import random
    from sda_amd import crypto
    from oracle import coracle
    rnd = random.Random(bits)
    for trial in range(3):
        for _ in range(100000):
            p = rnd.randrange(1 << (bits - 1), 1 << bits) | 1
            if _is_prime(p) and p > 64:
                break
        else:
            raise AssertionError("no prime found")
        k, t, n = rnd.choice([(3, 1, 8), (2, 2, 6), (3, 4, 8), (8, 2, 12), (1, 1, 3)])
        for _ in range(100000):
            w2, w3 = rnd.randrange(2, p), rnd.randrange(2, p)
            if len({pow(w2, e, p) for e in range(k + t + 1)}) == k + t + 1 and \
               len({1} | {pow(w3, j + 1, p) for j in range(n)}) == n + 1:
                break
        else:
            raise AssertionError("no suitable roots found")
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
        dim = rnd.choice([1, 17, 1000, 4097])
        B = (dim + k - 1) // k
        rng = np.random.default_rng(trial)
        P = 5
        secrets = rng.integers(-(1 << 63), (1 << 63) - 1, size=(P, dim), dtype=np.int64)
        rand = rng.integers(-(1 << 63), (1 << 63) - 1, size=(P, B * t), dtype=np.int64)
        gen = crypto.ShareGenerator(sch)
        shares = [gen.generate(secrets[i], rand[i]) for i in range(P)]
        for i in range(P):
            assert np.array_equal(shares[i], coracle.packed_generate(p, k, t, n, w2, w3, secrets[i], rand[i])), (p, k, t)
        comb = crypto.ShareCombiner(sch)
        sums = [comb.combine([shares[i][c] for i in range(P)]) for c in range(n)]
        subset = sorted(rnd.sample(range(n), k + t))
        got = crypto.SecretReconstructor(sch, dim).reconstruct([(c, sums[c]) for c in subset])
        want = [sum(int(secrets[i][j]) for i in range(P)) % p for j in range(dim)]
        assert got.tolist() == want
        # device CSPRNG path: reconstruct(sum of fresh sharings) is independent of the draws
        gen.set_drbg_key(KEY)
        shares2 = [gen.generate(secrets[i]) for i in range(P)]
        sums2 = [comb.combine([shares2[i][c] for i in range(P)]) for c in range(n)]
        assert crypto.SecretReconstructor(sch, dim).reconstruct([(c, sums2[c]) for c in subset]).tolist() == want
        rnd_draws = coracle.drbg_fill(KEY, 0, B, t, p) if t else np.zeros(0, dtype=np.int64)
        assert np.array_equal(shares2[0], coracle.packed_generate_csprng(p, k, t, n, w2, w3, secrets[0], rnd_draws, gen.csprng_share_map()))


def test_odd_strides_and_unaligned_bases_dev(gpu):
    """caller layouts that rule out 16-byte accesses (odd strides, 8-byte-aligned bases) take the scalar
    paths of the kernels; results must be identical"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(31)
    P, dim = 5, 1001
    for sch, k, t, n in ((crypto.PackedShamir(3, 8, 1, P62, W[8], W[9]), 3, 1, 8), (crypto.Additive(3, P62), 1, 2, 3)):
        B = (dim + k - 1) // k
        sec_stride, out_sp, out_sc = dim + 2, B + 1, P * (B + 1) + 3        # odd strides
        host = rng.integers(0, P62, size=(P, sec_stride), dtype=np.int64)
        d_sec = DeviceBuffer.from_numpy(np.concatenate([[0], host.reshape(-1)]))     # +1 element: base only 8-byte aligned
        rand = rng.integers(0, P62, size=(P, B * t), dtype=np.int64)
        d_rand = DeviceBuffer.from_numpy(rand)
        d_out = DeviceBuffer(n * out_sc + 8).zero()
        gen = crypto.ShareGenerator(sch)
        gen.generate_batch_dev(d_sec.at(1), P, dim, sec_stride, d_out.at(1), out_sp, out_sc, d_rand=d_rand.ptr, rand_stride=B * t)
        got = d_out.to_numpy()[1:]
        for p in range(P):
            want = (coracle.packed_generate(P62, k, t, n, W[8], W[9], host[p, :dim], rand[p]) if k == 3 else
                    coracle.additive_generate(P62, n, host[p, :dim], rand[p]))
            for j in range(n):
                assert np.array_equal(got[j * out_sc + p * out_sp: j * out_sc + p * out_sp + B], want[j]), (k, p, j)
        # clerk-sum straight from that odd-strided, misaligned layout
        comb = crypto.ShareCombiner(sch)
        d_sum = DeviceBuffer(n * B)
        comb.begin_dev(n, B)
        comb.update_dev(d_out.at(1), out_sc, P, out_sp)
        comb.finish_dev(d_sum.ptr)
        S = d_sum.to_numpy().reshape(n, B)
        for j in range(n):
            rows = np.stack([got[j * out_sc + p * out_sp: j * out_sc + p * out_sp + B] for p in range(P)])
            assert np.array_equal(S[j], coracle.combine(P62, rows))


@pytest.mark.parametrize("kind", ["packed_k3_t1_n8", "packed_k8_t2_n26", "additive_n3", "packed_odd_strides",
                                  "packed_k4_t3_n8", "packed_k6_t2_n8", "packed_k9_t6_n26",     # these three: run-time (k, t) form
                                  "packed_k8_t7_n26",                                           # limb GEMM on the matrix cores
                                  "packed_k40_t23_n242"])   # transform kernel: no dual-role form, clerk sum on the low-priority side stream
def test_dual_role_pipeline_equals_separate_launches(gpu, kind):
    """sda_share_generator_generate_combine_dev (tile i+1 generated while tile i is summed, one grid) must
    produce exactly the shares and clerk sums of generate_batch_dev + combiner update_dev."""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    dim, P, tiles = 40_000, 300, 3
    if kind == "additive_n3":
        sch, k, t, n = crypto.Additive(3, P62), 1, 2, 3
    elif kind == "packed_k8_t2_n26":
        k, t, n = 8, 2, 26
        sch = crypto.PackedShamir(k, n, t, P62, W[16], W[27])
    elif kind == "packed_k8_t7_n26":
        k, t, n = 8, 7, 26
        sch = crypto.PackedShamir(k, n, t, P62, W[16], W[27])
    elif kind == "packed_k4_t3_n8":
        k, t, n = 4, 3, 8
        sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
    elif kind == "packed_k6_t2_n8":
        k, t, n = 6, 2, 8
        sch = crypto.PackedShamir(k, n, t, P62, W[16], W[9])
    elif kind == "packed_k9_t6_n26":
        k, t, n = 9, 6, 26
        sch = crypto.PackedShamir(k, n, t, P62, W[16], W[27])
    elif kind == "packed_k40_t23_n242":
        k, t, n = 40, 23, 242
        sch = crypto.PackedShamir(k, n, t, P62, _root(P62, 64), _root(P62, 243))
    else:
        k, t, n = 3, 1, 8
        sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
    B = (dim + k - 1) // k
    Bs = (B + 15) // 16 * 16
    if kind == "packed_odd_strides":
        Bs = B + 1 if (B + 1) % 2 else B + 2          # odd row stride: the fused form is ruled out, fallback path
    secrets = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 77, P62, None))
    # reference: separate launches
    gen = crypto.ShareGenerator(sch); gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(sch)
    ref_shares = [DeviceBuffer(n * P * Bs).zero() for _ in range(tiles)]
    comb.begin_dev(n, B)
    for i in range(tiles):
        gen.generate_batch_dev(secrets.ptr, P, dim, dim, ref_shares[i].ptr, Bs, P * Bs, first_participant=i * P)
        comb.update_dev(ref_shares[i].ptr, P * Bs, P, Bs)
    ref_sums = DeviceBuffer(n * B)
    comb.finish_dev(ref_sums.ptr)
    # pipelined: K + 1 dual-role launches over two buffers
    gen2 = crypto.ShareGenerator(sch); gen2.set_drbg_key(KEY)
    comb2 = crypto.ShareCombiner(sch)
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    comb2.begin_dev(n, B)
    for i in range(tiles + 1):
        cur, prev = bufs[i % 2], bufs[(i - 1) % 2]
        gen2.generate_combine_dev(comb2, secrets.ptr, P if i < tiles else 0, dim, dim, cur.ptr, Bs, P * Bs,
                                  d_prev=prev.ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                  first_participant=i * P)
        if i < tiles:
            a = cur.to_numpy().reshape(n, P, Bs)[:, :, :B]
            b = ref_shares[i].to_numpy().reshape(n, P, Bs)[:, :, :B]
            assert np.array_equal(a, b), f"shares of tile {i}"
    sums = DeviceBuffer(n * B)
    comb2.finish_dev(sums.ptr)
    assert np.array_equal(sums.to_numpy(), ref_sums.to_numpy())
    # and against the oracle: clerk 0's sum over all tiles
    host = np.concatenate([r.to_numpy().reshape(n, P, Bs)[0, :, :B] for r in ref_shares])
    assert np.array_equal(sums.to_numpy()[:B], coracle.combine(P62, host))


def test_side_stream_pipeline_without_host_synchronisation(gpu):
    """the transform shape's generate_combine_dev forks the clerk sum of tile i-1 to a side stream and joins it before it
    returns control of `stream`: five launches back to back over two share buffers with NO host synchronisation in between
    (the buffer the next launch overwrites is the one the side stream is still reading unless the join works), against
    separate launches; also with the side stream switched off, and a clerk-sum grid smaller than the job"""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    k, t, n, dim, P, tiles = 40, 23, 242, 40 * 700 + 11, 96, 4
    sch = crypto.PackedShamir(k, n, t, P62, _root(P62, 64), _root(P62, 243))
    B = (dim + k - 1) // k
    Bs = (B + 15) // 16 * 16
    secrets = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 78, P62, None))
    gen = crypto.ShareGenerator(sch); gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(sch)
    buf = DeviceBuffer(n * P * Bs).zero()
    comb.begin_dev(n, B)
    for i in range(tiles):
        gen.generate_batch_dev(secrets.ptr, P, dim, dim, buf.ptr, Bs, P * Bs, first_participant=i * P)
        comb.update_dev(buf.ptr, P * Bs, P, Bs)
    want = DeviceBuffer(n * B)
    comb.finish_dev(want.ptr)
    want = want.to_numpy()
    for env in ({}, {"SDA_NO_SIDE_STREAM": "1"}):
        for kk, vv in env.items():
            set_knob(kk, vv)
        try:
            gen2 = crypto.ShareGenerator(sch); gen2.set_drbg_key(KEY)
            comb2 = crypto.ShareCombiner(sch)
            bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
            comb2.begin_dev(n, B)
            for i in range(tiles + 1):
                gen2.generate_combine_dev(comb2, secrets.ptr, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                          d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                          first_participant=i * P)
            got = DeviceBuffer(n * B)
            comb2.finish_dev(got.ptr)
            assert np.array_equal(got.to_numpy(), want), env
        finally:
            for kk in env:
                set_knob(kk, 0)


def test_device_entry_points_refuse_bad_arguments(gpu):
    """with a live device: wrong sizes, strides, states and alignments come back as status codes (and leave the
    library usable), they do not launch anything."""
    import ctypes as C
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBuffer
    lib = gpu
    bad, state = capi.ERR_INVALID_ARGUMENT, capi.ERR_STATE
    sch = crypto.PackedShamir(3, 8, 1, P62, W[8], W[9])
    gen, comb, codec = crypto.ShareGenerator(sch), crypto.ShareCombiner(sch), crypto.VarintCodec()
    buf = DeviceBuffer(4096).zero()
    st = DeviceBuffer(1).zero()
    g, c, v = gen._h, comb._h, codec._h
    # generate: NULL device pointers, empty work is fine
    assert lib.sda_share_generator_generate_batch_dev(g, None, 2, 10, 10, None, 0, 0, buf.ptr, 16, 64, None) == bad
    assert lib.sda_share_generator_generate_batch_dev(g, buf.ptr, 0, 10, 10, None, 0, 0, buf.ptr, 16, 64, None) == capi.OK
    # combiner: update / finish / pipelined step before begin
    assert lib.sda_share_combiner_update_dev(c, buf.ptr, 64, 2, 16, None) == state
    assert lib.sda_share_combiner_finish_dev(c, buf.ptr, None) == state
    assert lib.sda_share_generator_generate_combine_dev(g, c, buf.ptr, 1, 10, 10, 0, buf.ptr, 16, 64, None, 0, None) == state
    comb.begin_dev(8, 5)
    # pipelined step: the combiner must match the generator's (share_count, batches)
    assert lib.sda_share_generator_generate_combine_dev(g, c, buf.ptr, 1, 10, 10, 0, buf.ptr, 16, 64, None, 0, None) == bad
    assert lib.sda_share_generator_generate_combine_dev(g, c, buf.ptr, 1, 15, 15, 0, buf.ptr, 16, 64, None, 3, None) == bad   # d_prev NULL
    # wire format: rows not a multiple of the jobs, missing offsets, misaligned slots, short slots
    assert lib.sda_share_combiner_update_varint_dev(c, v, buf.ptr, 100, buf.ptr, 9, st.ptr, None) == bad
    assert lib.sda_share_combiner_update_varint_dev(c, v, buf.ptr, 100, None, 16, st.ptr, None) == bad
    assert lib.sda_share_combiner_update_varint_rows_dev(c, v, buf.ptr + 8, 64, buf.ptr, 8, st.ptr, None) == bad
    assert lib.sda_share_combiner_update_varint_rows_dev(c, v, buf.ptr, 40, buf.ptr, 8, st.ptr, None) == bad
    assert lib.sda_varint_encode_rows_dev(v, buf.ptr, 2, 10, 10, buf.ptr + 1024, 96, buf.ptr + 2048, None) == bad       # slot < 10 * len
    assert lib.sda_varint_encode_rows_dev(v, buf.ptr, 2, 10, 9, buf.ptr + 1024, 112, buf.ptr + 2048, None) == bad       # stride < len
    assert lib.sda_varint_decode_dev(v, buf.ptr, 100, None, 3, 4, buf.ptr + 1024, 4, st.ptr, None) == bad              # offsets needed
    assert lib.sda_varint_decode_rows_dev(v, buf.ptr, 24, buf.ptr + 512, 2, 2, buf.ptr + 1024, 2, st.ptr, None) == bad  # slot % 16
    # masking: strides, scheme mismatch
    mk = crypto.SecretMasker(crypto.Full(P62))
    assert lib.sda_secret_masker_mask_batch_dev(mk._h, buf.ptr, 2, 10, 9, 0, buf.ptr + 1024, 10, buf.ptr + 2048, 10, None) == bad
    mc = crypto.SecretMasker(crypto.ChaCha(P62, 10, 128))
    assert lib.sda_secret_masker_mask_batch_dev(mc._h, buf.ptr, 2, 9, 10, 0, buf.ptr + 1024, 4, buf.ptr + 2048, 10, None) == capi.ERR_ASSERTION
    assert lib.sda_secret_masker_mask_batch_dev(mc._h, buf.ptr, 2, 10, 10, 0, buf.ptr + 1024, 3, buf.ptr + 2048, 10, None) == bad
    # and the handles still work
    secrets = np.arange(10, dtype=np.int64)
    assert gen.generate(secrets).shape == (8, 4)


def test_bench_two_ranks_rccl_refusal_falls_back_loudly(gpu, tmp_path):
    """two ranks on ONE device with the library's RCCL communicator attempted: the 128-byte id travels over gloo, both
    ranks reach ncclCommInitRank, RCCL refuses the duplicate device ("invalid usage"), every rank takes the labelled
    host-staged exchange together, and the cross-rank result still verifies.  (On a node with one GPU per rank the same
    code path keeps the communicator.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDA_SHARE_GPU="try")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--participants", "120", "--dim", "65536", "--no-additional", "--no-cpu-baseline"]
    compact, line, out = _run_bench(cmd, tmp_path, env=env, timeout=600)
    assert line["n_gpus"] == 2 and line["verified_reconstruct_equals_sum"] is True
    assert compact["n_gpus"] == 2 and compact["rccl"]["unique_devices"] == 1 and "exchange_ms" in compact
    assert "RCCL" in line["config"]["exchange"]
    if "unavailable" in line["config"]["exchange"]:
        assert "sda_comm_init failed" in out.stderr
        assert line["rccl"]["ranks"] == 0 and line["rccl"]["path"].startswith("host-staged gloo (RCCL refused")
    assert line["rccl"]["unique_devices"] == 1


def test_bench_refused_communicator_is_fatal_without_the_rehearsal_switch(gpu):
    """What a real N-GPU node must do when the library's RCCL communicator cannot be set up: exit non-zero and print NO
    line.  Provoked here by pinning both ranks to device 0 (SDA_BENCH_DEVICE=0; RCCL refuses the duplicate device) WITHOUT
    SDA_SHARE_GPU, the only switch that allows the host-staged exchange."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "SDA_SHARE_GPU"}
    env["SDA_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--participants", "120", "--dim", "65536", "--no-additional", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode != 0
    assert "sda_comm_init failed" in out.stderr and "FATAL" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "a refused communicator must not print a result"


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--schedule", "serial", "--workload", "additive"]),
                                         (2, ["--workload", "packed26"])])          # config 4's shape (k=8, t=2, n=26)
def test_bench_multi_rank_rehearsal_on_one_gpu(gpu, ranks, extra, tmp_path):
    """bench.py's N > 1 path end to end - participant sharding, per-rank CSPRNG streams, the all-to-all / modular
    sum / all-gather exchange, max-over-ranks timing, and the cross-rank verification reconstruct(sum of every rank's
    clerk sums) == sum of every rank's secrets - with the ranks sharing this box's one GPU and gloo carrying the
    exchange through host memory (RCCL refuses two ranks on one device).  The kernels are the real ones."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDA_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "3",
           "--warmup", "1", "--participants", "120", "--dim", "65536", "--no-additional", "--no-cpu-baseline"] + extra
    compact, line, _ = _run_bench(cmd, tmp_path, env=env, timeout=600)
    assert line["n_gpus"] == ranks and line["verified_reconstruct_equals_sum"] is True
    assert compact["config"]["participants_total"] == ranks * 3 * 40 and compact["scaling"] == "weak"
    assert line["config"]["participants_total"] == ranks * 3 * 40 and line["scaling"] == "weak"
    assert f"{ranks * 120} participants" in line["config"]["workload"]                 # the label is what was processed
    assert line["rccl"] == dict(line["rccl"], ranks=0, unique_devices=1) and "host-staged" in line["rccl"]["path"]
    assert line["exchange_bytes_per_gpu"] == 8 * line["config"]["share_count"] * -(-65536 // line["config"]["secret_count"])


def test_bench_multi_gpu_legs_configs_4_and_5(gpu, tmp_path):
    """`bench.py --gpus N` (N > 1) as the driver launches it: after the config-3 line, BASELINE config 4 (packed Shamir
    t=2 k=8 n=26) and config 5 (k=3 t=1 n=8 + Lagrange reveal) run with the job's participants sharded over the ranks,
    each leg's clerk sums meeting in one modular reduce, each leg verified against the secrets of ALL ranks.  Rehearsed on
    the one GPU of this box (ranks share it, gloo carries the exchange) at a small job size and dimension."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDA_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--participants", "120", "--dim", "65536", "--no-cpu-baseline", "--leg-participants", "400", "--leg-dim", "98304"]
    compact, line, _ = _run_bench(cmd, tmp_path, env=env, timeout=900)
    assert line["verified_reconstruct_equals_sum"] is True and line["scaling"] == "weak"
    legs = line["additional_workloads"]
    assert set(legs) == {"config4_packed26", "config5_packed_dim16m"} == set(compact["additional_workloads"])
    assert all(v["verified"] is True and v["value"] > 0 for v in compact["additional_workloads"].values())
    assert compact["additional_workloads"]["config5_packed_dim16m"]["reveal_ms"] > 0
    c4, c5 = legs["config4_packed26"], legs["config5_packed_dim16m"]
    for leg, (k, t, n) in ((c4, (8, 2, 26)), (c5, (3, 1, 8))):
        cfg = leg["config"]
        assert (cfg["secret_count"], cfg["privacy_threshold"], cfg["share_count"]) == (k, t, n)
        assert leg["verified_reconstruct_equals_sum"] is True and leg["n_gpus"] == 2 and leg["value"] > 0
        assert cfg["participants_total"] == 400 and "200 per GPU" in cfg["job"]
        assert leg["exchange_bytes_per_gpu"] == 8 * n * -(-98304 // k)
        assert leg["roofline"]["bound"] in ("hbm", "valu") and leg["scaling"].startswith("strong")
    assert c5["reveal"]["dim"] == 98304 and c5["reveal"]["ms"] > 0


def test_bench_launches_its_own_ranks(gpu, tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (the form the driver used for its N = 1 record): the script
    starts its two ranks itself, stdout is exactly ONE JSON line with n_gpus 2, the `rccl` record, a `roofline` and a
    non-null `cpu_baseline` (rank 0's host cores, at any world size); the two multi-GPU legs are attached.  Ranks share
    this box's one GPU (SDA_SHARE_GPU=1).  Without the switch the same command is exit code 3 and prints no line: a
    one-GPU box cannot produce a two-GPU measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--participants", "120", "--dim", "65536",
            "--leg-participants", "400", "--leg-dim", "98304"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SDA_SHARE_GPU")}
    compact, line, _ = _run_bench([sys.executable, os.path.join(root, "bench.py")] + args, tmp_path, env=dict(env, SDA_SHARE_GPU="1"))
    assert compact["n_gpus"] == 2 and compact["cpu_baseline"]["value"] > 0 and compact["roofline"]["frac"] > 0
    assert line["n_gpus"] == 2 and line["verified_reconstruct_equals_sum"] is True and line["scaling"] == "weak"
    assert "WEAK" in line["scaling_note"] and "STRONG" in line["scaling_note"]
    assert line["rccl"]["unique_devices"] == 1 and line["roofline"]["frac"] > 0
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1
    assert set(line["additional_workloads"]) == {"config4_packed26", "config5_packed_dim16m"}
    for leg in line["additional_workloads"].values():
        assert leg["verified_reconstruct_equals_sum"] is True and leg["config"]["participants_total"] == 400
    refused = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                             env=env, cwd=root)
    assert refused.returncode == 3, (refused.returncode, refused.stderr[-1500:])
    assert "FATAL" in refused.stderr
    assert not [l for l in refused.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_exchange_watchdog_ends_a_run_whose_peer_never_arrives(gpu, tmp_path, launcher):
    """VERDICT r5 item 1 on the GPU box: two ranks (sharing this box's GPU), rank 1 falls asleep when it enters the WARM-UP
    EXCHANGE - the first modular reduce between the ranks, the call no multi-GPU run has ever made.  Rank 0 is then inside the
    exchange waiting for a peer that never arrives: its watchdog must end it after --exchange-timeout-s with exit code 5, a
    diagnosis (rank, device, peers, exchange path, RCCL version) on stderr and NO JSON line; the launcher - ours or
    torch.distributed.run - comes back promptly with a non-zero code."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SDA_SHARE_GPU="1", SDA_BENCH_TEST_HANG="warm-up exchange:1")
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--participants", "40", "--dim", "65536",
            "--no-cpu-baseline", "--no-additional", "--exchange-timeout-s", "10", "--deadline-s", "300",
            "--details", os.path.join(str(tmp_path), "d.json")]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", _free_port()] + args
    else:
        cmd = [sys.executable] + args
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    took = time.time() - t0
    assert out.returncode != 0 and (launcher == "torchrun" or out.returncode == 5), (out.returncode, out.stderr[-3000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout
    assert took < 240, took
    err = out.stderr
    assert "WATCHDOG: phase 'warm-up exchange: packed" in err and "did not finish within its limit of 10 s" in err
    assert "exiting with code 5, no JSON line" in err and "peers [" in err and "RCCL version" in err
    assert "SDA_BENCH_TEST_HANG: sleeping forever in phase 'warm-up exchange" in err
    for r in (0, 1):
        assert f"[bench] rank {r}/2 " in err


def test_bench_single_gpu_line_carries_configs_4_and_5_as_full_jobs(gpu, tmp_path):
    """N = 1: BASELINE configs 4 and 5 ride on the default line as `config4_full` / `config5_full` - the whole job on one
    GPU, streamed as resident tiles, verified against the column sums (here shrunk through --leg-participants / --leg-dim;
    the driver's run uses 1,000,000 and 100,000 participants)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--participants", "64", "--dim", "65536",
           "--no-cpu-baseline", "--leg-participants", "300", "--leg-dim", "98304"]
    compact, line, _ = _run_bench(cmd, tmp_path)
    legs = line["additional_workloads"]
    assert {"config4_full", "config5_full", "additive", "packed_pss728", "narrow_ref", "narrow26_ref", "narrow_pss728", "narrow_pss19682",
            "packed_tss_nodes", "packed_distinct", "additive_chacha12", "config4_chacha12"} == set(legs) == set(compact["additional_workloads"])
    assert all(v["verified"] is True for v in compact["additional_workloads"].values())
    # round 6: the distinct-inputs price and the round-count price ride on the line, labelled
    assert compact["additional_workloads"]["additive_chacha12"]["rounds"] == 12 == compact["additional_workloads"]["config4_chacha12"]["rounds"]
    assert legs["additive_chacha12"]["roofline"]["kernel"] == "fused_additive_kernel<12>" and "ChaCha12" in legs["additive_chacha12"]["config"]["randomness"]
    assert legs["config4_chacha12"]["config"]["share_count"] == 26 and legs["config4_chacha12"]["roofline"]["bound"] is None
    d = legs["packed_distinct"]
    assert d["config"]["inputs"].startswith("distinct") and d["fill_bytes_per_element"] == 8.0 and d["frac_with_fill"] == pytest.approx(d["frac_wall"] * 58.6667 / 50.6667, rel=1e-3)
    assert d["frac_wall"] == pytest.approx(d["path_roofline"]["frac_of_hbm_peak"]) and compact["additional_workloads"]["packed_distinct"]["frac"] == pytest.approx(d["frac_wall"], rel=1e-3)
    assert d["config"]["distinct_participants"] == d["config"]["participants_total"]
    assert compact["additional_workloads"]["packed_distinct"]["frac_with_fill"] == pytest.approx(d["frac_with_fill"], rel=1e-3)
    # the kernel names are the library's own report of what it launched (sda_debug_last_kernel)
    assert legs["narrow_ref"]["roofline"]["kernel"].startswith("fused_packed_n31_kernel<8, ") and legs["narrow_ref"]["verified_reconstruct_equals_sum"] is True
    assert legs["narrow_ref"]["config"]["library_path"] == "l31+n31" and legs["config4_full"]["roofline"]["kernel"] == "fused_packed_l31_kernel<8, 2, 20>"
    assert legs["additive"]["roofline"]["kernel"] == "fused_additive_kernel<20>"
    assert legs["packed_pss728"]["roofline"]["kernel"].startswith("packed_gen_fft_kernel<20, ") and "side stream" in legs["packed_pss728"]["roofline"]["kernel"]
    assert legs["narrow_pss728"]["config"]["modulus"] == 746497 and legs["narrow_pss728"]["verified_reconstruct_equals_sum"] is True
    # tss's two shipped parameter sets over tss's own primes run the limb GEMM on the matrix cores (dual-role launch)
    assert legs["narrow_pss728"]["roofline"]["kernel"] == "packed_gen_ngemm_kernel<4, 2>" and legs["narrow_pss728"]["config"]["library_path"] == "fft+ngemm"
    # the reference's own share map rides along: the headline is on the library's systematic map, this leg on tss's
    assert compact["config"]["csprng_share_map"].startswith("systematic (library)")
    assert legs["packed_tss_nodes"]["config"]["csprng_share_map"].startswith("tss nodes (reference)")
    assert legs["packed_tss_nodes"]["config"]["share_count"] == 8 and legs["packed_tss_nodes"]["verified_reconstruct_equals_sum"] is True
    assert legs["narrow_pss19682"]["config"]["share_count"] == 19682 and legs["narrow_pss19682"]["verified_reconstruct_equals_sum"] is True
    for key, (k, t, n) in (("config4_full", (8, 2, 26)), ("config5_full", (3, 1, 8))):
        cfg = legs[key]["config"]
        assert (cfg["secret_count"], cfg["privacy_threshold"], cfg["share_count"]) == (k, t, n)
        assert cfg["participants_total"] == 300 and cfg["dim"] == 98304 and "full job size on ONE GPU" in cfg["job"]
        assert legs[key]["verified_reconstruct_equals_sum"] is True
        assert legs[key]["roofline"]["traffic"] is None and legs[key]["roofline"]["traffic_note"].startswith("null")
    assert legs["config5_full"]["reveal"]["dim"] == 98304


def test_bench_distinct_inputs_mode(gpu, tmp_path):
    """--inputs distinct: every sub-tile shares different participants (tile i+1's secrets generated on a side stream
    while tile i runs); the result is verified against the column sums of ALL of them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--participants", "320", "--tile", "32",
           "--dim", "65536", "--no-cpu-baseline", "--no-additional", "--inputs", "distinct"]
    _, line, _ = _run_bench(cmd, tmp_path, timeout=600)
    assert line["verified_reconstruct_equals_sum"] is True
    assert line["config"]["distinct_participants"] == 320 == line["config"]["participants_total"]
    assert "320 distinct participants" in line["verified_against"]


@pytest.mark.parametrize("k,t,n,dim", [(8, 7, 26, 8 * 64 * 5 + 3), (8, 7, 26, 1), (8, 7, 26, 8 * 2048 + 8 * 77), (8, 2, 26, 8 * 300 + 1),
                                       (3, 4, 8, 3 * 1000 + 2), (3, 1, 8, 3 * 129), (8, 7, 26, 8 * 64),
                                       # shapes without a compiled instance: the run-time (k, t) form
                                       (9, 6, 26, 9 * 333 + 4), (13, 2, 26, 13 * 70), (1, 14, 26, 401), (2, 9, 26, 2 * 999 + 1),
                                       (5, 4, 26, 5 * 123), (16, 0, 26, 16 * 65 + 3)])
def test_limb_gemm_share_generation_vs_oracle(gpu, monkeypatch, k, t, n, dim):
    """packed_gen_mfma_kernel (the limb GEMM on the matrix cores; the default from k + t = 12, SDA_FORCE_MFMA=1 for every
    compiled shape) against the oracle's matrix form: injected randomness with any-i64 secrets (ragged last batch, partial
    64-batch steps, several workgroups per participant), then the device CSPRNG streams of three participants, and the
    round trip through reconstruct."""
    set_knob("SDA_FORCE_MFMA", "1")
    _share_gen_vs_oracle(k, t, n, dim)


def test_limb31_kernel_serves_8_7_26_when_the_limb_gemm_is_switched_off(gpu, monkeypatch):
    set_knob("SDA_NO_MFMA", "1")
    _share_gen_vs_oracle(8, 7, 26, 8 * 64 * 5 + 3)


@pytest.mark.parametrize("n", [15, 27, 31, 32, 79, 80, 81, 242])
def test_limb_gemm_clerk_counts(gpu, n):
    """odd and extreme clerk counts of the limb-GEMM kernel (its clerk loop alternates two accumulator sets; 242 clerks is
    what its constant table may hold): share points 3^1 .. 3^n.  (Up to 29 clerks the three-digit limb-31 kernel is the
    default for (8,7) since round 4: the knob keeps this test on the limb GEMM.)"""
    set_knob("SDA_FORCE_MFMA", 1)
    _share_gen_vs_oracle(8, 7, n, 8 * 200 + 5, w3=W[3], odd_stride=n in (27, 80))


@pytest.mark.parametrize("k,t,n,dim", [(8, 7, 26, 8 * 64 * 5 + 3), (8, 7, 27, 8 * 300 + 1), (8, 2, 26, 8 * 1000 + 7), (3, 4, 8, 3 * 2000 + 2),
                                       (8, 7, 15, 8 * 100), (3, 4, 7, 3 * 100 + 1)])
def test_three_digit_limb31_kernels_vs_oracle(gpu, k, t, n, dim):
    """the shapes compiled in the three-digit form (R = 2^93: groups of seven terms, carry normalisation, one reduction per dot
    product) through their DEFAULT path over the 62-bit prime: injected any-i64 randomness, device CSPRNG, reconstruct"""
    _share_gen_vs_oracle(k, t, n, dim, w3=W[3] if n not in (8, 26) else None)
    # adversarial operands: secrets and injected draws whose balanced 31-bit limbs are all extreme (+-2^30) and all alike within
    # a batch, so that every product of a column has the same sign and the largest magnitude the limbs allow
    from sda_amd import crypto
    from oracle import coracle
    w2 = _root(P62, k + t + 1)
    w3 = W[3] if n not in (8, 26) else _root(P62, n + 1)
    gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, P62, w2, w3))
    ext = []
    for a in (-(1 << 30), (1 << 30) - 1):
        for b in (-(1 << 30) + 1, (1 << 30) - 1, -(1 << 30)):
            x = b * (1 << 31) + a
            if abs(x) <= (P62 - 1) // 2:
                ext.append(x % P62)
    ext += [0, 1, P62 - 1, (P62 - 1) // 2, (P62 + 1) // 2]
    B = len(ext) * len(ext)
    secrets = np.array([ext[(b // len(ext))] for b in range(B) for _ in range(k)], dtype=np.int64)
    rand = np.array([ext[(b % len(ext))] for b in range(B) for _ in range(t)], dtype=np.int64)
    assert np.array_equal(gen.generate(secrets, rand), coracle.packed_generate(P62, k, t, n, w2, w3, secrets, rand))


def _share_gen_vs_oracle(k, t, n, dim, w3=None, odd_stride=False):
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    w2, w3 = _root(P62, k + t + 1) if (k + t + 1) & (k + t) == 0 else None, w3 or _root(P62, n + 1)
    if w2 is None:                                     # k + t + 1 not a power of two: any root of a power-of-two order >= k + t + 1
        m2 = 1
        while m2 < k + t + 1:
            m2 *= 2
        w2 = _root(P62, m2)
    rng = np.random.default_rng(k * 131 + t * 7 + dim)
    sch = crypto.PackedShamir(k, n, t, P62, w2, w3)
    gen = crypto.ShareGenerator(sch)
    B = gen.batch_count(dim)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    rand = rng.integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
    got = gen.generate(secrets, rand)
    want = coracle.packed_generate(P62, k, t, n, w2, w3, secrets, rand)
    assert np.array_equal(got, want)
    gen.set_drbg_key(KEY)
    P = 3
    sec2 = rng.integers(0, P62, size=(P, dim), dtype=np.int64)
    sec2[0, : min(dim, 5)] = [0, P62 - 1, 1, P62 // 2, P62 // 2 + 1][: min(dim, 5)]
    d_sec = DeviceBuffer.from_numpy(sec2)
    Bs = (B | 1) if odd_stride else B + (B & 1)        # odd row stride: 8-byte aligned rows only
    d_out = DeviceBuffer(P * n * Bs).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=(1 << 40) + 5)
    out = d_out.to_numpy().reshape(P, n, Bs)
    for q in range(P):
        w = coracle.packed_generate_csprng(P62, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, (1 << 40) + 5 + q, B, t, P62),
                                           gen.csprng_share_map())
        assert np.array_equal(out[q, :, :B], w), f"participant {q}"
    idx = sorted(rng.choice(n, size=t + k, replace=False).tolist())
    rec = crypto.SecretReconstructor(sch, dim).reconstruct([(i, out[1, i, :B]) for i in idx])
    assert np.array_equal(rec, sec2[1])
