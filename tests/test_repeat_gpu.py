"""-m gpu: every share-generation kernel family, the same launch many times over, every element against the restatement.

Round 5 met a defect in the limb GEMM that a single comparison passes nine times out of ten (a store that now and then left with
the next row's value, tests/test_ngemm_gpu.py::test_narrow_limb_gemm_repeated_launches).  Nothing like it has been seen in the
other families - this file is where it would show: dimensions that fill every CU several times over (so that workgroups compete
for the memory pipeline), ragged tails, three participants, the library's own randomness, 25 launches per shape."""
import numpy as np
import pytest

from test_parity_gpu import KEY, P62, W

pytestmark = pytest.mark.gpu

P31 = 2147482801                                  # the largest prime = 1 mod 432 below 2^31 (the n31 kernels; bench.py uses it too)


def _root(p, order):
    for g in range(2, 5000):
        w = pow(g, (p - 1) // order, p)
        if (p - 1) % order == 0 and all(pow(w, order // f, p) != 1 for f in (2, 3) if order % f == 0):
            return w
    raise AssertionError("no root")


def capi_last_kernel():
    from sda_amd import capi
    return capi.load().sda_debug_last_kernel().decode()


def _root62(order):
    g = next(g for g in range(2, 500) if all(pow(g, (P62 - 1) // f, P62) != 1 for f in (2, 3)))
    return pow(g, (P62 - 1) // order, P62)


SHAPES = [
    # name, modulus, (k, t, n, omega_secrets, omega_shares) or additive n, elements, kernel family the library must report
    ("config 3", P62, (3, 1, 8, W[8], W[9]), 3 * 300_001 + 1, "packed_gen_l31_kernel"),
    ("config 4", P62, (8, 2, 26, W[16], W[27]), 8 * 100_003, "packed_gen_l31_kernel"),
    ("(8,7,26)", P62, (8, 7, 26, W[16], W[27]), 8 * 60_001 + 5, "packed_gen_l31_kernel"),
    ("(12,3,26): 62-bit limb GEMM", P62, (12, 3, 26, W[16], W[27]), 12 * 40_001 + 7, "packed_gen_mfma_kernel"),
    ("(40,23,242): transform", P62, (40, 23, 242, "o64", "o243"), 40 * 3_001 + 9, "packed_gen_fft_kernel"),
    ("(20,13,80): matrix in global memory", P62, (20, 13, 80, 3, 5), 20 * 20_001 + 3, "packed_gen_l31"),
    ("(40,30,100): any-shape kernel", P62, (40, 30, 100, 3, 5), 40 * 4_001 + 3, "packed_gen"),
    ("(3,4,8) over a 31-bit prime", P31, (3, 4, 8, "o8", "o9"), 3 * 200_003, "packed_gen_n31_kernel"),
    ("(8,7,26) over a 31-bit prime", P31, (8, 7, 26, "o16", "o27"), 8 * 60_001, "packed_gen_n31_kernel"),
    ("additive, 3 shares", P62, 3, 700_001, "additive_gen_kernel"),
    ("additive over 433", 433, 3, 700_001, "additive_gen_kernel"),
]


@pytest.mark.parametrize("name,p,shape,dim,kernel", SHAPES, ids=[s[0] for s in SHAPES])
def test_repeated_launches_every_element(gpu, name, p, shape, dim, kernel):
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    P, first = 3, 77
    rng = np.random.default_rng(dim)
    sec = rng.integers(0, p, size=(P, dim), dtype=np.int64)
    if isinstance(shape, tuple):
        k, t, n, w2, w3 = shape
        if isinstance(w2, str):                                    # "o64": an element of that order
            w2, w3 = ((_root62(int(w2[1:])), _root62(int(w3[1:]))) if p == P62 else (_root(p, int(w2[1:])), _root(p, int(w3[1:]))))
        sch = crypto.PackedShamir(k, n, t, p, w2, w3)
    else:
        n, t, k = shape, shape - 1, 1
        sch = crypto.Additive(n, p)
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    B = gen.batch_count(dim) if isinstance(shape, tuple) else dim
    Bs = (B + 15) // 16 * 16
    if isinstance(shape, tuple):
        want = np.stack([coracle.packed_generate_csprng(p, k, t, n, w2, w3, sec[q], coracle.drbg_fill(KEY, first + q, B, t, p), gen.csprng_share_map())
                         for q in range(P)])
    else:
        draws = [coracle.drbg_fill(KEY, first + q, B, t, p).reshape(B, t).T for q in range(P)]
        want = np.stack([np.vstack([d, np.mod(sec[q].astype(object) - d.astype(object).sum(axis=0), p).astype(np.int64)]) for q, d in enumerate(draws)])
    d_sec = DeviceBuffer.from_numpy(sec)
    d_out = DeviceBuffer(P * n * Bs)
    for rep in range(25):
        d_out.zero()
        gen.generate_batch_dev(d_sec.ptr, P, dim, dim, d_out.ptr, n * Bs, Bs, first_participant=first)
        got = d_out.to_numpy().reshape(P, n, Bs)
        bad = np.argwhere(got[:, :, :B] != want)
        assert len(bad) == 0, (name, rep, len(bad), bad[:4].tolist())
        assert not got[:, :, B:].any()
    assert kernel.split("_kernel")[0] in capi_last_kernel(), capi_last_kernel()
