"""-m gpu parity tests of the narrow limb GEMM (sda_amd/csrc/ngemm_kernels.hip): packed-Shamir share generation of the large
tss-valid shapes over a prime below 2^23 (tss's own 746497 and 5038849) on the matrix cores.  Every case is checked against the
oracle (packed_shamir.rs:42 -> tss share(), restated in oracle/) with injected randomness and with the device CSPRNG, and
against the transform kernel serving the same handle parameters (knob SDA_NO_NGEMM, include/sda_hip_debug.h): all three agree
bit for bit."""
import numpy as np
import pytest

from conftest import set_knob

pytestmark = pytest.mark.gpu

KEY = bytes((i * 11 + 3) & 0xFF for i in range(32))
TSS_P1, TSS_P2 = 746497, 5038849


def capi_last_kernel():
    from sda_amd import capi
    return capi.load().sda_debug_last_kernel().decode()


def _root(p, order):
    assert (p - 1) % order == 0
    for g in range(2, 2000):
        w = pow(g, (p - 1) // order, p)
        if all(pow(w, order // f, p) != 1 for f in (2, 3) if order % f == 0):
            return w
    raise AssertionError("no root")


CASES = [
    (TSS_P1, 100, 155, 728, 95660, 610121, 100 * 150 + 37),          # tss's PSS_155_728_100, its own roots: 4 steps, 151 batches (two workgroups, ragged)
    (TSS_P1, 100, 155, 728, 95660, 610121, 1),                       # a single secret
    (TSS_P2, 100, 155, 19682, 4318906, 1814687, 250),                # tss's PSS_155_19682_100: 1231 row tiles
    (TSS_P1, 40, 23, 242, None, None, 40 * 300 + 1),                 # 63 terms: one step, 256 batches per workgroup, 301 batches
    (TSS_P1, 70, 57, 242, None, None, 70 * 260),                     # 127 terms: two steps
    (TSS_P1, 300, 211, 728, None, None, 300 * 70 + 11),              # 511 terms: eight steps, 64 batches per workgroup
    (TSS_P2, 20, 11, 80, None, None, 20 * 100 + 3),                  # 31 terms, n = 80: five row tiles
    (TSS_P1, 9, 22, 242, None, None, 9 * 129),                       # draws start inside the first step; n = 242: the last row tile ragged
    (TSS_P1, 20, 13, 50, "o64", "o81", 20 * 77 + 3),                 # NOT tss-valid (34 of 64 nodes, 50 of 80 share points): no transform kernel
    (TSS_P1, 33, 0, 80, "o64", "o81", 33 * 50 + 1),                  # no privacy threshold: no draws, no systematic map
]


@pytest.mark.parametrize("p,k,t,n,w2,w3,dim", CASES)
def test_narrow_limb_gemm_vs_oracle_and_transform(gpu, p, k, t, n, w2, w3, dim):
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    tss_valid = not isinstance(w2, str)
    w2 = _root(p, int(w2[1:])) if isinstance(w2, str) else w2 or _root(p, k + t + 1)
    w3 = _root(p, int(w3[1:])) if isinstance(w3, str) else w3 or _root(p, n + 1)
    rng = np.random.default_rng(k * 1000 + n)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    secrets[: min(dim, 6)] = [0, p - 1, p, -1, p // 2, p // 2 + 1][: min(dim, 6)]
    P = 3
    sec2 = rng.integers(0, p, size=(P, dim), dtype=np.int64)
    results = []
    for gemm in (True, False):
        set_knob("SDA_NO_NGEMM", 0 if gemm else 1)
        if k + t <= 32 and tss_valid:
            set_knob("SDA_FORCE_FFT", 1)                               # small tss-valid shapes default to the matrix kernels
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
        gen = crypto.ShareGenerator(sch)
        # the limb GEMM's draws are shares 0..t-1 (systematic map, like every matrix-form kernel); the transform kernel's are tss's nodes
        assert gen.csprng_share_map() == (gen.SHARE_MAP_SYSTEMATIC if (gemm or not tss_valid) and t > 0 else gen.SHARE_MAP_TSS_NODES)
        B = gen.batch_count(dim)
        rand = np.random.default_rng(7).integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
        got = gen.generate(secrets, rand)
        assert np.array_equal(got, coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)), gemm
        gen.set_drbg_key(KEY)
        d_sec = DeviceBuffer.from_numpy(sec2)
        Bs = (B + 15) // 16 * 16 + 16
        first = (1 << 33) + 9
        for share_map in ((gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES) if (gemm or not tss_valid) and t > 0 else (gen.SHARE_MAP_TSS_NODES,)):
            gen.set_csprng_share_map(share_map)
            d_out = DeviceBuffer(P * n * Bs).zero()
            gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=first)
            o = d_out.to_numpy().reshape(P, n, Bs)
            for q in range(P):
                w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, first + q, B, t, p), share_map)
                assert np.array_equal(o[q, :, :B], w), (gemm, share_map, q)
            assert not o[:, :, B:].any()                               # nothing beyond a row's batches
        results.append((got, o.copy()))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])


def test_narrow_limb_gemm_odd_strides_and_clerk_major_layout(gpu):
    """job-major output ([clerk][participant][batch]) with an odd row stride and an unaligned base: the 8-byte stores need no
    alignment; nothing is written past a row"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, k, t, n = TSS_P1, 40, 23, 242
    w2, w3 = _root(p, 64), _root(p, 243)
    dim, P, first = 40 * 77 + 5, 2, 123
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    Bs = B | 1
    rng = np.random.default_rng(5)
    stride = dim + 3
    sec = rng.integers(-(1 << 62), 1 << 62, size=(P, stride), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    d_out = DeviceBuffer(n * P * Bs + 1).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, stride, d_out.ptr + 8, Bs, P * Bs, first_participant=first)
    out = d_out.to_numpy()[1:].reshape(n, P, Bs)
    for q in range(P):
        want = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q, :dim], coracle.drbg_fill(KEY, first + q, B, t, p),
                                              gen.csprng_share_map())
        assert np.array_equal(out[:, q, :B], want), q
    assert not out[:, :, B:].any()


@pytest.mark.parametrize("form", ["far", "unaligned", "aligned"])
def test_narrow_limb_gemm_store_paths(gpu, form):
    """the three ways a share leaves the limb GEMM (ngemm_kernels.hip, finish_whole / finish_masked / finish_far), each with whole
    workgroups (256 batches) AND a ragged last one, a ragged last row tile (n - t = 13) and two participants:
    aligned - 16-byte buffer stores, two adjacent batch columns per lane; unaligned (base + 8 bytes, odd strides) - 8-byte buffer
    stores under masks; far - clerk rows 256 MiB apart (more than a buffer descriptor's 32-bit offsets reach in 16 rows): plain
    64-bit addresses.  Nothing may be written outside the rows' first B columns."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, k, t, n = TSS_P1, 10, 7, 20
    w2, w3 = _root(p, 32), _root(p, 27)
    dim, P, first = 10 * 600 + 3, 2, 9
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    assert B == 601
    Bs = B + 7 if form != "unaligned" else B + 6                    # 608 (even) / 607 (odd)
    stride_clerk = {"far": (1 << 25) + 2, "unaligned": P * Bs + 1, "aligned": P * Bs}[form]
    shift = 1 if form == "unaligned" else 0
    rng = np.random.default_rng(17)
    sec = rng.integers(-(1 << 62), 1 << 62, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    d_out = DeviceBuffer(n * stride_clerk + shift).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr + 8 * shift, Bs, stride_clerk, first_participant=first)
    assert capi_last_kernel().startswith("packed_gen_ngemm_kernel")
    rows = np.stack([d_out.to_numpy(P * Bs, shift + j * stride_clerk) for j in range(n)]).reshape(n, P, Bs)
    for q in range(P):
        want = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], coracle.drbg_fill(KEY, first + q, B, t, p), gen.csprng_share_map())
        assert np.array_equal(rows[:, q, :B], want), q
    assert not rows[:, :, B:].any()
    if form == "unaligned":                                         # the element between two clerk rows, and the one in front of the base
        assert not any(d_out.to_numpy(1, shift + j * stride_clerk + P * Bs)[0] for j in range(n - 1)) and d_out.to_numpy(1)[0] == 0


@pytest.mark.parametrize("k,t,n,dim", [(70, 57, 242, 70 * 260), (40, 23, 242, 52001), (100, 155, 728, 26001)])
def test_narrow_limb_gemm_repeated_launches(gpu, k, t, n, dim):
    """the same shares 40 times over (KS = 2, 1 and 4; whole workgroups and a ragged one; three participants).  Round 5 met a store
    that now and then left with the NEXT row's value in 16 of its lanes - one launch in ten, only in the waves that go from the
    reduction straight into the next row tile's products (ngemm_kernels.hip, finish_whole): a single comparison per shape passes
    nine times out of ten"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, P, first = TSS_P1, 3, (1 << 33) + 9
    w2, w3 = _root(p, {127: 128, 63: 64, 255: 256}[k + t]), _root(p, {242: 243, 728: 729}[n])
    gen = crypto.ShareGenerator(crypto.PackedShamir(k, n, t, p, w2, w3))
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16 + 16
    sec = np.random.default_rng(k + n).integers(0, p, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    want = np.stack([coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], coracle.drbg_fill(KEY, first + q, B, t, p), gen.csprng_share_map())
                     for q in range(P)])
    d_out = DeviceBuffer(P * n * Bs)
    for rep in range(40):
        d_out.zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=first)
        got = d_out.to_numpy().reshape(P, n, Bs)[:, :, :B]
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (rep, len(bad), bad[:4].tolist())
    assert capi_last_kernel().startswith("packed_gen_ngemm_kernel")


def test_narrow_limb_gemm_share_combine_reveal_roundtrip(gpu):
    """tss's PSS_155_728_100 over tss's prime through the pipelined step (the dual-role launch: share generation of tile i and the
    clerk sum of tile i - 1 in one grid), clerk sums against the oracle, then the reveal from an arbitrary t + k clerks = the sum
    of the secrets"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    p, k, t, n, w2, w3 = TSS_P1, 100, 155, 728, 95660, 610121
    dim, P, tiles = 100 * 40 + 7, 5, 3
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    from oracle import coracle
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(sch)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    rng = np.random.default_rng(11)
    secs = [rng.integers(0, p, size=(P, dim), dtype=np.int64) for _ in range(tiles)]
    d_secs = [DeviceBuffer.from_numpy(s) for s in secs]
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    comb.begin_dev(n, B)
    for i in range(tiles + 1):
        gen.generate_combine_dev(comb, d_secs[i].ptr if i < tiles else 0, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                 d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                 first_participant=i * P)
    d_sums = DeviceBuffer(n * B)
    comb.finish_dev(d_sums.ptr)
    sums = d_sums.to_numpy().reshape(n, B)
    # every clerk's sum against the oracle's combine over the oracle's shares of all 15 participants (dual-role launch: the clerk
    # sum of tile i - 1 runs in workgroups of the grid that generates tile i)
    want = [coracle.packed_generate_csprng(p, k, t, n, w2, w3, secs[i][q], coracle.drbg_fill(KEY, i * P + q, B, t, p), gen.csprng_share_map())
            for i in range(tiles) for q in range(P)]
    for c in (0, 1, t - 1, t, n // 2, n - 1):
        assert np.array_equal(sums[c], coracle.combine(p, np.stack([wq[c] for wq in want]))), c
    idx = sorted(rng.choice(n, size=t + k, replace=False).tolist())
    rec = crypto.SecretReconstructor(sch, dim).reconstruct([(i, sums[i]) for i in idx])
    truth = np.zeros(dim, dtype=np.int64)
    for s in secs:
        truth = (truth + s.sum(axis=0)) % p
    assert np.array_equal(rec, truth)


def test_narrow_limb_gemm_repeated_dual_role_launches(gpu):
    """the repeated-launch check of above through the pipelined step: the shares of a tile are only ever seen by the clerk sum
    (dual-role grid), so every clerk's sum of 9 participants is compared, 15 times over (KS = 2: the shape that exposed the
    store that left with the next row's value)"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, k, t, n = TSS_P1, 70, 57, 242
    w2, w3 = _root(p, 128), _root(p, 243)
    dim, P, tiles = 70 * 260, 3, 3
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    rng = np.random.default_rng(23)
    secs = [rng.integers(0, p, size=(P, dim), dtype=np.int64) for _ in range(tiles)]
    d_secs = [DeviceBuffer.from_numpy(s) for s in secs]
    shares = [coracle.packed_generate_csprng(p, k, t, n, w2, w3, secs[i][q], coracle.drbg_fill(KEY, i * P + q, B, t, p), gen.csprng_share_map())
              for i in range(tiles) for q in range(P)]
    want = np.stack([coracle.combine(p, np.stack([sh[c] for sh in shares])) for c in range(n)])
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    d_sums = DeviceBuffer(n * B)
    for rep in range(15):
        comb = crypto.ShareCombiner(sch)
        comb.begin_dev(n, B)
        for i in range(tiles + 1):
            gen.generate_combine_dev(comb, d_secs[i].ptr if i < tiles else 0, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                     d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                     first_participant=i * P)
            if rep == 0 and i == 1:                     # a call that carries both roles: ONE launch
                assert gpu.sda_debug_last_kernel().decode() == "packed_gen_ngemm_kernel<2, 2>"
        comb.finish_dev(d_sums.ptr)
        got = d_sums.to_numpy().reshape(n, B)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (rep, len(bad), bad[:4].tolist())


@pytest.mark.parametrize("form,P,batches", [("waves", 23, 300), ("waves", 41, 38), ("workgroups", 23, 300), ("waves", 5, 261)])
def test_narrow_limb_gemm_clerk_waves_and_clerk_workgroups(gpu, form, P, batches):
    """the two forms of the dual-role launch (round 6).  "waves": three clerk waves inside every share-generation workgroup sum the
    previous tile with two register sets of ten rows in flight (ng_clerk_wave: hand-written counted waits), and
    ngemm_clerk_rest_kernel sums what they did not get to - P = 23 / 41 rows make quanta of 10 + 10 + 3 / 4 x 10 + 1 (the partial
    last quantum, an item that ends mid-workgroup, the recorded (item, row) the follow-up kernel resumes from); 38 batches leave
    most lanes of a 128-column item without a column.  "workgroups": rounds 4 - 5's clerk workgroups in the grid, now persistent
    (knob SDA_NGEMM_CLERK_WG).  An ODD number of batches (261) cannot use the clerk waves' 16-byte accesses to the running sums
    and takes the workgroup form by itself.  Every clerk's sum over three tiles is compared with the oracle, five times over."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    if form == "workgroups":
        set_knob("SDA_NGEMM_CLERK_WG", 1)
    p, k, t, n = TSS_P1, 70, 57, 242
    w2, w3 = _root(p, 128), _root(p, 243)
    dim, tiles = 70 * batches - 11, 3
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    assert B == batches
    Bs = (B + 15) // 16 * 16
    rng = np.random.default_rng(29)
    secs = [rng.integers(0, p, size=(P, dim), dtype=np.int64) for _ in range(tiles)]
    d_secs = [DeviceBuffer.from_numpy(s) for s in secs]
    want = np.zeros((n, B), dtype=np.int64)
    for i in range(tiles):
        for q in range(P):
            sh = coracle.packed_generate_csprng(p, k, t, n, w2, w3, secs[i][q], coracle.drbg_fill(KEY, i * P + q, B, t, p), gen.csprng_share_map())
            want = (want + sh) % p
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    d_sums = DeviceBuffer(n * B)
    for rep in range(5):
        comb = crypto.ShareCombiner(sch)
        comb.begin_dev(n, B)
        for i in range(tiles + 1):
            gen.generate_combine_dev(comb, d_secs[i].ptr if i < tiles else 0, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                     d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                     first_participant=i * P)
            if i == 1:
                assert gpu.sda_debug_last_kernel().decode() == "packed_gen_ngemm_kernel<2, 2>"      # ONE share-generation launch carries both roles
        comb.finish_dev(d_sums.ptr)
        got = d_sums.to_numpy().reshape(n, B)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (form, rep, len(bad), bad[:4].tolist())


def test_narrow_limb_gemm_rejected_draw_pairs_are_redone_from_the_retry_stream(gpu):
    """The paired rule rejects a candidate word with probability (2^64 mod p^2) / 2^64 - below 2^-18, so rare that a parity test of
    ordinary size never sees one.  The limb GEMM notes rejected pairs in a mask and redoes them in a cold function called LAST
    (ng_draw_fixup: round 6 - a call to the retry stream inside the loop made the pass save 27 registers in scratch memory on every
    call).  Here the prime is the one below 2^23 with the highest rejection rate among those that admit the shape (8211457:
    2.95e-6) and the job has 12 million draw pairs: 35 rejections expected (none at all: probability e^-35).  With the library's
    own randomness the draws ARE shares 0..t-1, so those rows are compared with the oracle's draws, every one of them."""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, k, t, n = 8211457, 40, 23, 242
    assert p <= 0x7F7F7F and (p - 1) % 64 == 0 and (p - 1) % 243 == 0
    w2, w3 = _root(p, 64), _root(p, 243)
    P, B = 2, 500_000
    dim = k * B
    expected_rejections = ((1 << 64) % (p * p)) / 2.0 ** 64 * P * B * ((t + 1) // 2)
    assert expected_rejections > 30, expected_rejections
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    assert gen.path_name().endswith("ngemm") and gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC and gen.batch_count(dim) == B
    secrets = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 77, p, None))
    out = DeviceBuffer(n * P * B)
    gen.generate_batch_dev(secrets.ptr, P, dim, dim, out.ptr, B, P * B, first_participant=5)
    assert gpu.sda_debug_last_kernel().decode() == "packed_gen_ngemm_kernel<1, 4>"
    for q in range(P):
        want = coracle.drbg_fill(KEY, 5 + q, B, t, p).reshape(B, t)
        for i in range(t):
            got = out.to_numpy(B, (i * P + q) * B)
            bad = np.flatnonzero(got != want[:, i])
            assert len(bad) == 0, (q, i, len(bad), bad[:4].tolist())


def test_narrow_limb_gemm_fallbacks_keep_the_shares(gpu):
    """what surrounds the kernel: (1) another ChaCha round count (A/B only) has no limb-GEMM instance - on this tss-valid shape the
    transform kernel serves those calls, which is tss's map whatever was requested, and csprng_share_map() SAYS so (round 4
    silently sent the systematic request to the any-shape kernel with every draw of the tile materialised); (2) a clerk-major layout with an odd
    row stride rules the dual-role launch out (its clerk role reads with 16-byte loads) - the two ordinary launches give the sums"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    p, k, t, n = TSS_P1, 40, 23, 242
    w2, w3 = _root(p, 64), _root(p, 243)
    dim, P, first = 40 * 50 + 3, 2, 5
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    rng = np.random.default_rng(3)
    sec = rng.integers(0, p, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    gen.set_drbg_rounds(12)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    assert gen.path_name() == "fft+ngemm"
    for asked in (gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES):
        gen.set_csprng_share_map(asked)
        share_map = gen.csprng_share_map()
        assert share_map == gen.SHARE_MAP_TSS_NODES                     # the map the NEXT call uses, not the one asked for
        d_out = DeviceBuffer(P * n * Bs).zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=first)
        o = d_out.to_numpy().reshape(P, n, Bs)
        for q in range(P):
            draws = coracle.drbg_fill(KEY, first + q, B, t, p, rounds=12)
            assert np.array_equal(o[q, :, :B], coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], draws, share_map)), (share_map, q)
        assert capi_last_kernel().startswith("packed_gen_fft_kernel<12, ")
    gen.set_drbg_rounds(20)                                             # back under ChaCha20 the systematic request holds again
    gen.set_csprng_share_map(gen.SHARE_MAP_SYSTEMATIC)
    assert gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC
    # (2) odd clerk-major strides through generate_combine_dev
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(sch)
    Bo = B | 1
    bufs = [DeviceBuffer(n * P * Bo + 1).zero() for _ in range(2)]
    comb.begin_dev(n, B)
    tiles = 2
    for i in range(tiles + 1):
        gen.generate_combine_dev(comb, d_sec.ptr if i < tiles else 0, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr + 8, Bo, P * Bo,
                                 d_prev=bufs[(i - 1) % 2].ptr + 8 if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                 first_participant=first + i * P)
    d_sums = DeviceBuffer(n * B)
    comb.finish_dev(d_sums.ptr)
    sums = d_sums.to_numpy().reshape(n, B)
    want = [coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], coracle.drbg_fill(KEY, first + i * P + q, B, t, p), gen.csprng_share_map())
            for i in range(tiles) for q in range(P)]
    for c in (0, t - 1, t, n - 1):
        assert np.array_equal(sums[c], coracle.combine(p, np.stack([wq[c] for wq in want]))), c


def test_narrow_limb_gemm_random_shapes(gpu):
    """24 random (k, t, n, dimension) over tss's two primes - term counts on both sides of every 64-term step boundary, row counts
    on both sides of the 16-row tiles, batch counts around the workgroup sizes - each against the oracle: injected randomness and
    the device CSPRNG under both share maps"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(20260930)
    for trial in range(24):
        p = (TSS_P1, TSS_P2)[trial % 2]
        o2, o3 = (1024, 729) if p == TSS_P1 else (256, 19683)
        kt = int(rng.choice([17, 31, 63, 64, 65, 100, 127, 128, 129, 200, 255])) if trial < 20 else int(rng.integers(257, 512))
        kt = min(kt, o2 - 1)
        t = int(rng.integers(0, kt))
        k = kt - t
        n = int(rng.integers(kt, min(o3 - 1, kt + 300) + 1))
        w2, w3 = _root(p, o2), _root(p, o3)
        B = int(rng.choice([1, 7, 63, 64, 65, 127, 129, 255, 257, 300]))
        dim = max(1, B * k - int(rng.integers(0, k)))
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
        gen = crypto.ShareGenerator(sch)
        assert gen.batch_count(dim) == B
        secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
        rand = rng.integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
        assert np.array_equal(gen.generate(secrets, rand), coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)), (trial, p, k, t, n, dim)
        gen.set_drbg_key(KEY)
        sec = rng.integers(0, p, size=(1, dim), dtype=np.int64)
        d_sec = DeviceBuffer.from_numpy(sec)
        Bs = B + 3
        for share_map in ((gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES) if t > 0 else (gen.SHARE_MAP_TSS_NODES,)):
            gen.set_csprng_share_map(share_map)
            d_out = DeviceBuffer(n * Bs).zero()
            gen.generate_batch_dev(d_sec.ptr, 1, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=trial)
            o = d_out.to_numpy().reshape(n, Bs)
            w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[0], coracle.drbg_fill(KEY, trial, B, t, p), share_map)
            assert np.array_equal(o[:, :B], w), (trial, p, k, t, n, dim, share_map)
            assert not o[:, B:].any()


def test_narrow_limb_gemm_dual_role_random_shapes(gpu):
    """the dual-role launch over random shapes: every instance (1, 2, 4, 8 steps), 1 - 45 participants per tile (the clerk waves'
    quanta of ten rows: fewer rows than one quantum, exact multiples, ragged ends), batch counts that fill a 128-column clerk item
    partly, exactly or several times, even (clerk waves) and odd (clerk workgroups), row strides with padding.  The shares
    themselves are pinned by the tests above; here the clerk sums of three tiles are compared with the column sums of the tiles
    as they lie in device memory - every element."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    rng = np.random.default_rng(606)
    for trial in range(14):
        p = (TSS_P1, TSS_P2)[trial % 2]
        o2, o3 = (1024, 729) if p == TSS_P1 else (256, 19683)
        kt = min(int(rng.choice([20, 63, 64, 100, 128, 200, 255, 300])), o2 - 1)
        t = int(rng.integers(1, kt))
        k = kt - t
        n = int(rng.integers(kt, kt + 120))
        w2, w3 = _root(p, o2), _root(p, o3)
        B = int(rng.choice([2, 64, 126, 128, 130, 256, 258, 300, 514, 65, 257]))
        P = int(rng.choice([1, 3, 9, 10, 11, 20, 29, 30, 45]))
        dim, tiles = B * k - int(rng.integers(0, k)), 3
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
        gen = crypto.ShareGenerator(sch)
        gen.set_drbg_key(KEY)
        assert gen.batch_count(dim) == B
        Bs = (B + 15) // 16 * 16 + 16 * (trial % 2)
        secs = [DeviceBuffer.from_numpy(rng.integers(0, p, size=(P, dim), dtype=np.int64)) for _ in range(tiles)]
        bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
        comb = crypto.ShareCombiner(sch)
        comb.begin_dev(n, B)
        want = np.zeros((n, B), dtype=np.int64)
        for i in range(tiles + 1):
            gen.generate_combine_dev(comb, secs[i].ptr if i < tiles else 0, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                     d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                     first_participant=i * P)
            if i < tiles:
                tile = bufs[i % 2].to_numpy().reshape(n, P, Bs)[:, :, :B]
                assert tile.min() >= 0 and tile.max() < p
                want = (want + tile.sum(axis=1)) % p
        d_sums = DeviceBuffer(n * B)
        comb.finish_dev(d_sums.ptr)
        got = d_sums.to_numpy().reshape(n, B)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (trial, p, k, t, n, B, P, Bs, len(bad), bad[:4].tolist())
