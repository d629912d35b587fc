"""-m gpu parity tests of the narrow-modulus kernels (p < 2^31: sda_amd/csrc/narrow_gen.inc.hpp; p < 2^30: the uint32_t
instantiation of the transform kernel) - the reference's own valid domain for packed Shamir (tss multiplies i64 residues
without widening: p = 433 of full_loop.rs:57-64, tss's shipped 746497 and 5038849).  Every case is checked against the
oracle AND against the 62-bit kernels serving the same prime (knob SDA_NO_NARROW, include/sda_hip_debug.h)."""
import numpy as np
import pytest

from conftest import set_knob

pytestmark = pytest.mark.gpu

KEY = bytes((i * 7 + 1) & 0xFF for i in range(32))
TSS_P1, TSS_P2 = 746497, 5038849
P31, P30, P29 = 2147482801, 1073738161, 536868433          # largest primes = 1 mod 432 below 2^31, 2^30, 2^29


def _root(p, order):
    assert (p - 1) % order == 0
    for g in range(2, 2000):
        w = pow(g, (p - 1) // order, p)
        if all(pow(w, order // f, p) != 1 for f in (2, 3) if order % f == 0):
            return w
    raise AssertionError("no root")


def _pow2_at_least(x):
    m = 1
    while m < x:
        m *= 2
    return m


def _check_shape(p, k, t, n, dim, narrow, dual_role=True):
    """injected randomness (any-i64 inputs) and the device CSPRNG (both share maps) against the oracle; reconstruct round
    trip; the dual-role launch against separate launches"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    if not narrow:
        set_knob("SDA_NO_NARROW", 1)
    w2, w3 = _root(p, _pow2_at_least(k + t + 1)), _root(p, n + 1)
    rng = np.random.default_rng(p % 1000 + k * 131 + t * 7 + dim)
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    B = gen.batch_count(dim)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    secrets[: min(dim, 6)] = [0, p - 1, p, -1, p // 2, p // 2 + 1][: min(dim, 6)]
    rand = rng.integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
    got = gen.generate(secrets, rand)
    assert np.array_equal(got, coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand))
    gen.set_drbg_key(KEY)
    P, first = 3, (1 << 40) + 5
    sec2 = rng.integers(0, p, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec2)
    Bs = (B + 15) // 16 * 16
    out = {}
    for share_map in (gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES):
        if t == 0 and share_map == gen.SHARE_MAP_SYSTEMATIC:
            continue
        gen.set_csprng_share_map(share_map)
        d_out = DeviceBuffer(n * P * Bs).zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, Bs, P * Bs, first_participant=first)      # job-major [n][P][Bs]
        o = d_out.to_numpy().reshape(n, P, Bs)
        for q in range(P):
            w = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, first + q, B, t, p), share_map)
            assert np.array_equal(o[:, q, :B], w), (share_map, q)
        assert not o[:, :, B:].any()
        out[share_map] = o
    o = out[gen.SHARE_MAP_TSS_NODES]
    idx = sorted(rng.choice(n, size=t + k, replace=False).tolist())
    rec = crypto.SecretReconstructor(sch, dim).reconstruct([(i, o[i, 1, :B]) for i in idx])
    assert np.array_equal(rec, sec2[1])
    if dual_role and t > 0:
        gen.set_csprng_share_map(gen.SHARE_MAP_SYSTEMATIC)
        comb = crypto.ShareCombiner(sch)
        comb.begin_dev(n, B)
        bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
        tiles = 2
        for i in range(tiles + 1):
            gen.generate_combine_dev(comb, d_sec.ptr, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                     d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                     first_participant=first + i * P)
        d_sums = DeviceBuffer(n * B)
        comb.finish_dev(d_sums.ptr)
        sums = d_sums.to_numpy().reshape(n, B)
        tile0 = out[gen.SHARE_MAP_SYSTEMATIC][:, :, :B]
        tile1 = np.stack([coracle.packed_generate_systematic(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, first + P + q, B, t, p))
                          for q in range(P)], axis=1)
        for c in range(n):
            assert np.array_equal(sums[c], coracle.combine(p, np.concatenate([tile0[c], tile1[c]])))
    return got


SHAPES = [(433, 3, 4, 8, 3 * 1000 + 2), (433, 3, 4, 8, 1), (TSS_P1, 3, 4, 8, 3 * 700), (TSS_P1, 8, 7, 26, 8 * 300 + 5),
          (TSS_P2, 8, 7, 26, 8 * 513), (TSS_P2, 8, 2, 26, 8 * 300 + 1), (P29, 8, 7, 26, 8 * 257 + 3), (P30, 8, 7, 26, 8 * 129),
          (P31, 8, 7, 26, 8 * 300 + 7), (P31, 3, 4, 8, 3 * 999 + 1), (P31, 3, 1, 8, 3 * 640), (P31, 1, 1, 2, 777), (P31, 4, 0, 8, 4 * 100 + 1),
          (TSS_P2, 13, 3, 26, 13 * 40), (TSS_P1, 2, 9, 26, 2 * 500 + 1), (P31, 5, 4, 26, 5 * 123)]


@pytest.mark.parametrize("p,k,t,n,dim", SHAPES)
def test_narrow_matrix_kernels_vs_oracle_and_wide(gpu, p, k, t, n, dim):
    a = _check_shape(p, k, t, n, dim, narrow=True)
    b = _check_shape(p, k, t, n, dim, narrow=False)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("p,k,t,n,w2,w3,dim", [
    (TSS_P1, 100, 155, 728, 95660, 610121, 100 * 29 + 37),            # tss's PSS_155_728_100 with its own roots
    (TSS_P2, 100, 155, 19682, 4318906, 1814687, 250),                 # tss's PSS_155_19682_100 (one batch per workgroup)
    (TSS_P1, 40, 23, 242, None, None, 40 * 40),                       # 64 / 243 points
    (P29, 7, 8, 26, None, None, 7 * 100 + 3),                         # forced below: small tss-valid shape, 29-bit prime
    (P30, 3, 4, 8, None, None, 3 * 50)])                              # 2^30 <= p: stays on the 64-bit transform kernel
def test_narrow_transform_kernel_vs_oracle_and_wide(gpu, p, k, t, n, w2, w3, dim):
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    w2 = w2 or _root(p, k + t + 1)
    w3 = w3 or _root(p, n + 1)
    rng = np.random.default_rng(k + n)
    secrets = rng.integers(-(1 << 62), 1 << 62, size=dim, dtype=np.int64)
    sec2 = rng.integers(0, p, size=(2, dim), dtype=np.int64)
    results = []
    for narrow in (True, "no_lazy", False):                                 # lazy radix-3 levels (default where (4b+4)p < 2^32), reduced, wide
        set_knob("SDA_NO_NGEMM", 1)                                         # the transform kernel itself (tests/test_ngemm_gpu.py: the limb GEMM)
        if k + t <= 32:
            set_knob("SDA_FORCE_FFT", 1)
        set_knob("SDA_NO_NARROW", 0 if narrow else 1)
        set_knob("SDA_NO_LAZY", 1 if narrow == "no_lazy" else 0)
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
        gen = crypto.ShareGenerator(sch)
        assert gen.csprng_share_map() == gen.SHARE_MAP_TSS_NODES           # the transform kernel's draws are tss's nodes
        B = gen.batch_count(dim)
        rand = np.random.default_rng(7).integers(-(1 << 62), 1 << 62, size=B * t, dtype=np.int64)
        got = gen.generate(secrets, rand)
        assert np.array_equal(got, coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)), narrow
        gen.set_drbg_key(KEY)
        d_sec = DeviceBuffer.from_numpy(sec2)
        Bs = B + (B & 1)
        d_out = DeviceBuffer(2 * n * Bs).zero()
        gen.generate_batch_dev(d_sec.ptr, 2, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=70)
        o = d_out.to_numpy().reshape(2, n, Bs)
        for q in range(2):
            w = coracle.packed_generate(p, k, t, n, w2, w3, sec2[q], coracle.drbg_fill(KEY, 70 + q, B, t, p))
            assert np.array_equal(o[q, :, :B], w), (narrow, q)
        results.append(o.copy())
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])


@pytest.mark.parametrize("p,k,t,n,rows", [(433, 3, 4, 8, 7), (433, 3, 4, 8, 8), (TSS_P1, 8, 7, 26, 15), (TSS_P2, 8, 7, 26, 16),
                                          (P31, 8, 7, 26, 15), (P31, 8, 7, 26, 20), (P31, 3, 1, 8, 4), (P29, 3, 1, 8, 6)])
def test_narrow_reveal_vs_oracle_and_wide(gpu, p, k, t, n, rows):
    """packed_reconstruct_n31_kernel (the reveal over a narrow prime; up to 16 clerk rows, more fall back to the 64-bit
    kernel) against the oracle's per-batch Newton interpolation (batched.rs:68-97, packed_shamir.rs:73-77) and against the
    64-bit kernel, for an arbitrary clerk subset and any-i64 share values"""
    from sda_amd import crypto
    from oracle import coracle
    w2, w3 = _root(p, _pow2_at_least(k + t + 1)), _root(p, n + 1)
    rng = np.random.default_rng(rows * 17 + k)
    dim = k * 1000 + 1
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    B = (dim + k - 1) // k
    idx = sorted(rng.choice(n, size=rows, replace=False).tolist())
    secrets = rng.integers(0, p, size=dim, dtype=np.int64)
    rand = rng.integers(0, p, size=B * t, dtype=np.int64)
    shares = coracle.packed_generate(p, k, t, n, w2, w3, secrets, rand)
    signed = shares[idx].copy()
    signed[::2] -= p                                             # the reference's representatives live in (-p, p)
    want = coracle.packed_reconstruct(p, k, t, w2, w3, dim, idx, shares[idx])
    assert np.array_equal(want, secrets)
    got = crypto.SecretReconstructor(sch, dim).reconstruct([(i, signed[j]) for j, i in enumerate(idx)])
    assert np.array_equal(got, want)
    set_knob("SDA_NO_NARROW", 1)
    wide = crypto.SecretReconstructor(sch, dim).reconstruct([(i, signed[j]) for j, i in enumerate(idx)])
    assert np.array_equal(wide, want)


@pytest.mark.parametrize("p,k,t,n", [(P31, 8, 7, 26), (TSS_P1, 3, 4, 8), (433, 3, 4, 8)])
def test_narrow_kernels_with_odd_strides_and_unaligned_rows(gpu, p, k, t, n):
    """caller layouts that rule out 16-byte stores (odd clerk stride, odd participant stride, secrets on an 8-byte boundary):
    the narrow kernels take their scalar paths, results identical to the oracle; nothing is written past a row"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    w2, w3 = _root(p, _pow2_at_least(k + t + 1)), _root(p, n + 1)
    rng = np.random.default_rng(n + k)
    dim, P, first = k * 333 + 2, 3, 9
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim)
    stride = dim + 3
    sec = rng.integers(-(1 << 62), 1 << 62, size=(P, stride), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    Bs = B | 1                                                    # odd row stride
    d_out = DeviceBuffer(P * n * Bs + 1).zero()
    gen.generate_batch_dev(d_sec.ptr, P, dim, stride, d_out.ptr + 8, n * Bs, Bs, first_participant=first)   # base on an 8-byte boundary
    out = d_out.to_numpy()[1:].reshape(P, n, Bs)
    for q in range(P):
        want = coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q, :dim], coracle.drbg_fill(KEY, first + q, B, t, p),
                                              gen.csprng_share_map())
        assert np.array_equal(out[q, :, :B], want), q
    assert not out[:, :, B:].any()


@pytest.mark.parametrize("p", [P31, 4611686006577364993])
def test_share_count_equal_to_the_reconstruction_threshold(gpu, p):
    """n = t + k (every share is needed; the systematic map leaves n - t = k interpolated rows): 7 of the 8 non-trivial ninth
    roots of unity as share points, over a narrow and over the 62-bit prime - generate (both share maps), reconstruct from
    all seven clerks, and one clerk too few is refused with the reference's error (packed_shamir.rs:75)"""
    from sda_amd import capi, crypto
    from oracle import coracle
    k, t, n = 3, 4, 7
    g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    w2, w3 = pow(g, (p - 1) // 8, p), pow(g, (p - 1) // 9, p)
    sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    rng = np.random.default_rng(3)
    dim = 3 * 500 + 1
    B = gen.batch_count(dim)
    secrets = rng.integers(0, p, size=dim, dtype=np.int64)
    for share_map in (gen.SHARE_MAP_SYSTEMATIC, gen.SHARE_MAP_TSS_NODES):
        gen.set_csprng_share_map(share_map)
        stream = 0 if share_map == gen.SHARE_MAP_SYSTEMATIC else 1
        got = gen.generate(secrets)
        want = coracle.packed_generate_csprng(p, k, t, n, w2, w3, secrets, coracle.drbg_fill(KEY, stream, B, t, p), share_map)
        assert np.array_equal(got, want), share_map
        rec = crypto.SecretReconstructor(sch, dim)
        assert np.array_equal(rec.reconstruct([(c, got[c]) for c in range(n)]), secrets)
        with pytest.raises(capi.SdaError) as e:
            rec.reconstruct([(c, got[c]) for c in range(n - 1)])
        assert e.value.code == capi.ERR_NOT_ENOUGH_SHARES
