"""Round 5: hazards removed rather than documented, and the library reporting what it ran.

* the share combiner orders its own accumulators across streams (include/sda_hip.h "Conventions"): generate_combine_dev on
  stream A followed by finish_dev on stream B returns the complete clerk sums - for the dual-role kernels and for the
  transform shapes whose clerk sum runs on the generator's side stream (the reference's caller, clerk.rs:80-86, has one
  thread and no streams; it cannot be asked to keep a stream rule);
* SDA_VALUES_RUST_SIGNED share generation with the library's randomness draws inside the kernel (additive.rs:42-47 with
  Rust's `%`): no participants x len x (n - 1) scratch, same draws as the canonical mode;
* sda_debug_last_kernel / sda_share_generator_path_name say what ran."""
import os
import sys

import numpy as np
import pytest

from conftest import set_knob

pytestmark = pytest.mark.gpu


class Stream:
    """a non-blocking HIP stream from the library's test-only helpers (no second HIP binding in the test process)"""

    def __init__(self):
        import ctypes as C
        from sda_amd import capi
        self._lib, self._h = capi.hooks_library(), C.c_void_p()       # stateless helpers of libsda_hip_test.so; the handles stay in the release library
        capi.check(self._lib.sda_debug_stream_create(C.byref(self._h)))
        self.cuda_stream = self._h.value

    def synchronize(self):
        from sda_amd import capi
        capi.check(self._lib.sda_debug_stream_synchronize(self._h))

    def __del__(self):
        try:
            self._lib.sda_debug_stream_destroy(self._h)
        except Exception:
            pass


def mem_free():
    import ctypes as C
    from sda_amd import capi
    f, t = C.c_size_t(), C.c_size_t()
    capi.check(capi.hooks_library().sda_debug_mem_info(C.byref(f), C.byref(t)))
    return f.value

P62 = 4611686006577364993
W = {8: 631229665360524489, 9: 3451275676410824977, 16: 2589100645267092065, 27: 365137883145458390}
KEY = bytes((i * 5 + 9) & 0xFF for i in range(32))


@pytest.mark.parametrize("form", ["dual_role_l31", "side_stream_transform", "two_launches_generic", "additive"])
def test_combiner_orders_its_accumulators_across_streams(gpu, form):
    """generate_combine_dev on stream A, finish_dev on stream B, 50 times in a row: the oracle's sums every time.  Nothing but
    the library orders B after A (no event, no synchronisation between the two calls on the caller's side)."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    A, Bst = Stream(), Stream()
    if form == "additive":
        sch, n, k, t = crypto.Additive(3, P62), 3, 1, 2
    else:
        n, k, t = 8, 3, 4
        sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
        if form == "side_stream_transform":
            set_knob("SDA_FORCE_FFT", 1)
        if form == "two_launches_generic":
            set_knob("SDA_FORCE_GENERIC", 1)
    P, dim, tiles = 48, k * 40000 + 2, 2        # even: the dual-role kernels read the secrets with 16-byte loads
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    assert gen.path_name() == {"dual_role_l31": "l31", "side_stream_transform": "fft", "two_launches_generic": "generic", "additive": "additive"}[form]
    comb = crypto.ShareCombiner(sch)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    secrets = coracle.fill_synthetic(P, dim, 0, 99, P62)
    d_sec = DeviceBuffer.from_numpy(secrets)
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    d_sums = DeviceBuffer(n * B)
    # the oracle's sums of both tiles (tile i uses streams i * P .. i * P + P - 1)
    want = np.zeros((n, B), dtype=object)
    rows = [[] for _ in range(n)]
    for i in range(tiles):
        for p in range(P):
            draws = coracle.drbg_fill(KEY, i * P + p, B, t, P62)
            sh = (coracle.additive_generate(P62, n, secrets[p], draws) if form == "additive" else
                  coracle.packed_generate_csprng(P62, k, t, n, W[8], W[9], secrets[p], draws, gen.csprng_share_map()))
            for c in range(n):
                rows[c].append(sh[c])
    want = np.stack([coracle.combine(P62, np.stack(rows[c])) for c in range(n)])
    from sda_amd import capi
    for rep in range(50):
        comb.begin_dev(n, B, stream=A.cuda_stream)
        for i in range(tiles + 1):
            gen.generate_combine_dev(comb, d_sec.ptr, P if i < tiles else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                     d_prev=bufs[(i - 1) % 2].ptr if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                     first_participant=i * P, stream=A.cuda_stream)
            if rep == 0 and i == 1:
                name = capi.load().sda_debug_last_kernel().decode()
                assert {"dual_role_l31": name == "fused_packed_l31_kernel<3, 4, 20>",
                        "side_stream_transform": name.startswith("packed_gen_fft_kernel<20, ") and name.endswith("combine_update_walk_kernel (side stream)"),
                        "two_launches_generic": name == "packed_gen_generic_kernel + combine_update_kernel (two launches)",
                        "additive": name == "fused_additive_kernel<20>"}[form], name
        comb.finish_dev(d_sums.ptr, stream=Bst.cuda_stream)                # ANOTHER stream, nothing in between
        Bst.synchronize()
        got = np.empty(n * B, dtype=np.int64)
        capi.check(capi.load().sda_dev_download(got.ctypes.data, d_sums.ptr, n * B * 8))
        assert np.array_equal(got.reshape(n, B), want), (form, rep)
    capi.check(capi.load().sda_dev_synchronize())


def test_update_dev_on_another_stream_after_the_side_stream_sum(gpu):
    """the same for update_dev: tile sums issued by generate_combine_dev (side-stream clerk sum) on stream A, one more tile
    added with update_dev on stream B, finish on the default stream"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    set_knob("SDA_FORCE_FFT", 1)
    A, Bst = Stream(), Stream()
    n, k, t = 8, 3, 4
    sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
    P, dim = 32, 3 * 30000
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(sch)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    secrets = coracle.fill_synthetic(P, dim, 7, 5, P62)
    d_sec = DeviceBuffer.from_numpy(secrets)
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    extra = np.random.default_rng(1).integers(0, P62, size=(n, 5, Bs), dtype=np.int64)
    d_extra = DeviceBuffer.from_numpy(extra)
    rows = [[] for _ in range(n)]
    for p in range(P):
        sh = coracle.packed_generate(P62, k, t, n, W[8], W[9], secrets[p], coracle.drbg_fill(KEY, p, B, t, P62))   # transform = tss's map
        for c in range(n):
            rows[c].append(sh[c])
    want = np.stack([coracle.combine(P62, np.stack(rows[c] + [extra[c, r, :B] for r in range(5)])) for c in range(n)])
    d_sums = DeviceBuffer(n * B)
    for rep in range(20):
        comb.begin_dev(n, B, stream=Bst.cuda_stream)
        for i in range(2):
            gen.generate_combine_dev(comb, d_sec.ptr, P if i == 0 else 0, dim, dim, bufs[i].ptr, Bs, P * Bs, d_prev=bufs[0].ptr if i else 0,
                                     prev_participants=P if i else 0, first_participant=0, stream=A.cuda_stream)
        comb.update_dev(d_extra.ptr, 5 * Bs, 5, Bs, stream=Bst.cuda_stream)
        comb.finish_dev(d_sums.ptr)
        assert np.array_equal(d_sums.to_numpy().reshape(n, B), want), rep


@pytest.mark.parametrize("q", [433, P62])
def test_signed_share_generation_draws_inside_the_kernel(gpu, q):
    """SDA_VALUES_RUST_SIGNED without injected randomness: shares 0..n-2 are the sda-drbg-v1 draws of the canonical kernel
    (same streams, same indexing), the last share is the reference's fold (acc - r) % q with Rust's `%` on the RAW secret
    (additive.rs:42-47) - bit for bit against the oracle's rust_signed restatement on those draws, any i64 secrets"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle, pyoracle as po
    n, P, dim, first = 4, 3, 1001, 11
    sch = crypto.Additive(n, q)
    rng = np.random.default_rng(q % 97)
    secrets = rng.integers(-(1 << 62) + 1, (1 << 62) - 1, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(secrets)
    outs = {}
    for mode in ("canonical", "rust_signed"):
        gen = crypto.ShareGenerator(sch).set_value_mode(mode)
        gen.set_drbg_key(KEY)
        d_out = DeviceBuffer(P * n * dim).zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * dim, dim, first_participant=first)
        outs[mode] = d_out.to_numpy().reshape(P, n, dim)
    from sda_amd import capi
    assert capi.load().sda_debug_last_kernel().decode() == "signed_additive_gen_drbg_kernel<20>"
    osch = po.AdditiveSecretSharing(n, q, "rust_signed")
    for p in range(P):
        draws = coracle.drbg_fill(KEY, first + p, dim, n - 1, q)
        want = po.generate(osch, [int(v) for v in secrets[p]], [int(v) for v in draws])
        assert [list(map(int, r)) for r in outs["rust_signed"][p]] == want, p
        # the two modes agree modulo q, and their draws are identical
        assert np.array_equal(outs["canonical"][p][:n - 1], outs["rust_signed"][p][:n - 1])
        assert np.array_equal(outs["canonical"][p][n - 1], np.array([int(v) % q for v in outs["rust_signed"][p][n - 1]], dtype=np.int64))


def test_signed_share_generation_of_a_bench_sized_tile_needs_no_scratch(gpu):
    """2000 participants x 1 Mi x n = 3 in the reference's representatives: round 4 materialised every draw first (33.5 GB for
    this tile, 117 GB at n = 8); now the device memory in use grows by less than 1 GB across the call"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle, pyoracle as po
    from sda_amd import capi
    lib = capi.load()
    n, P, dim = 3, 2000, 1 << 20
    sch = crypto.Additive(n, P62)
    gen = crypto.ShareGenerator(sch).set_value_mode("rust_signed")
    gen.set_drbg_key(KEY)
    d_sec = DeviceBuffer(P * dim)
    capi.check(lib.sda_fill_synthetic_dev(d_sec.ptr, P, dim, dim, 0, 0x5DA5, P62, None))
    d_out = DeviceBuffer(n * P * dim)
    capi.check(lib.sda_dev_synchronize())
    free0 = mem_free()
    gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, dim, P * dim, first_participant=0)      # job-major [n][P][dim]
    capi.check(lib.sda_dev_synchronize())
    free1 = mem_free()
    assert free0 - free1 < (1 << 30), (free0 - free1) / 2**30
    osch = po.AdditiveSecretSharing(n, P62, "rust_signed")
    for p in (0, 1999):
        sec = coracle.fill_synthetic(1, dim, p, 0x5DA5, P62)[0][:4096]
        draws = coracle.drbg_fill(KEY, p, dim, n - 1, P62)[:4096 * (n - 1)]
        want = po.generate(osch, [int(v) for v in sec], [int(v) for v in draws])
        for c in range(n):
            got = np.empty(4096, dtype=np.int64)
            capi.check(lib.sda_dev_download(got.ctypes.data, d_out.ptr + 8 * ((c * P + p) * dim), 4096 * 8))
            assert list(map(int, got)) == want[c], (p, c)


def test_the_library_reports_the_kernels_it_ran(gpu):
    from sda_amd import capi, crypto
    lib = capi.load()
    last = lambda: lib.sda_debug_last_kernel().decode()
    g = crypto.ShareGenerator(crypto.PackedShamir(3, 8, 1, P62, W[8], W[9]))
    g.generate(np.arange(30, dtype=np.int64))
    assert g.path_name() == "l31" and last() == "packed_gen_l31_kernel<3, 1, 20, true>", last()
    g = crypto.ShareGenerator(crypto.PackedShamir(8, 26, 2, P62, W[16], W[27]))
    g.generate(np.arange(80, dtype=np.int64))
    assert last() == "packed_gen_l31_kernel<8, 2, 20, true>"
    g = crypto.ShareGenerator(crypto.PackedShamir(3, 8, 4, 433, 354, 150))
    g.generate(np.arange(30, dtype=np.int64))
    assert g.path_name() == "l31+n31" and last().startswith("packed_gen_n31_kernel<8, 16, 20>"), last()
    g = crypto.ShareGenerator(crypto.Additive(3, 433))
    g.generate(np.arange(10, dtype=np.int64))
    assert g.path_name() == "additive" and last() == "additive_gen_kernel<20, true>", last()
    assert len(lib.sda_build_id()) == 16


@pytest.mark.parametrize("wide", ["karatsuba", "plain", False])
def test_config4_dot_product_as_one_group_and_as_seven_plus_three(gpu, wide):
    """BASELINE config 4's (8,2,26): the 10-term dot product as ONE three-digit group in its Karatsuba form (round 6, three
    multiply-adds per term: the default where the host admits it on the constants of both share maps), in the plain one-group form
    (knob SDA_NO_KARATSUBA) and in the 7 + 3 grouping (knob SDA_NO_WIDE_GROUP) - all against the oracle, on random and on
    adversarial operands (every balanced limb +-2^30, alike within a batch: the Karatsuba form's middle column wraps there),
    injected randomness and the device CSPRNG, separate and dual-role launches"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    if not wide:
        set_knob("SDA_NO_WIDE_GROUP", 1)
    elif wide == "plain":
        set_knob("SDA_NO_KARATSUBA", 1)
    k, t, n = 8, 2, 26
    sch = crypto.PackedShamir(k, n, t, P62, W[16], W[27])
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    ext = []
    for a in (-(1 << 30), (1 << 30) - 1):
        for b in (-(1 << 30) + 1, (1 << 30) - 1, -(1 << 30)):
            x = b * (1 << 31) + a
            if abs(x) <= (P62 - 1) // 2:
                ext.append(x % P62)
    ext += [0, 1, P62 - 1, (P62 - 1) // 2, (P62 + 1) // 2]
    Bx = len(ext) * len(ext)
    secrets = np.array([ext[(b // len(ext))] for b in range(Bx) for _ in range(k)], dtype=np.int64)
    rand = np.array([ext[(b % len(ext))] for b in range(Bx) for _ in range(t)], dtype=np.int64)
    assert np.array_equal(gen.generate(secrets, rand), coracle.packed_generate(P62, k, t, n, W[16], W[27], secrets, rand))
    rng = np.random.default_rng(5)
    P, dim = 6, k * 777 + 6
    sec = rng.integers(-(1 << 62), 1 << 62, size=(P, dim), dtype=np.int64)
    d_sec = DeviceBuffer.from_numpy(sec)
    B = gen.batch_count(dim)
    Bs = (B + 15) // 16 * 16
    bufs = [DeviceBuffer(n * P * Bs).zero() for _ in range(2)]
    comb = crypto.ShareCombiner(sch)
    comb.begin_dev(n, B)
    for i in range(3):
        gen.generate_combine_dev(comb, d_sec.ptr, P if i < 2 else 0, dim, dim, bufs[i % 2].ptr, Bs, P * Bs,
                                 d_prev=bufs[(i - 1) % 2].ptr if i else 0, prev_participants=P if i else 0, first_participant=i * P)
    from sda_amd import capi
    assert capi.load().sda_debug_last_kernel().decode() == "fused_packed_l31_kernel<8, 2, 20>"
    d_sums = DeviceBuffer(n * B)
    comb.finish_dev(d_sums.ptr)
    sums = d_sums.to_numpy().reshape(n, B)
    want = [coracle.packed_generate_csprng(P62, k, t, n, W[16], W[27], sec[q], coracle.drbg_fill(KEY, i * P + q, B, t, P62), gen.csprng_share_map())
            for i in range(2) for q in range(P)]
    tile1 = bufs[1].to_numpy().reshape(n, P, Bs)
    for q in range(P):
        assert np.array_equal(tile1[:, q, :B], want[P + q]), q
    for c in range(n):
        assert np.array_equal(sums[c], coracle.combine(P62, np.stack([w[c] for w in want]))), c
