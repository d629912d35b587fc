#!/usr/bin/env python3
"""Share-gen rate of the transform kernel for a few tss-valid shapes beyond PSS_155_728_100 (62-bit prime, dim 1 Mi):
the other shape tss ships (PSS_155_19682_100: one batch per workgroup, twiddles from global memory, 197 share values per
secret - bound by the WRITES), n + 1 = 3^7 (one batch per workgroup, twiddles in LDS) and two small ones.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sda_amd import capi, crypto
from sda_amd.device import DeviceBuffer, synchronize
P62 = 4611686006577364993                 # p - 1 = 2^16 * 3^8 * ...: roots of order up to 65536 and 6561
P62B = 4611686018374987777                # the largest prime below 2^62 with 256 * 19683 | p - 1 (for n + 1 = 3^9)


def root(p, order):
    assert (p - 1) % order == 0, (p, order)
    g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
    w = pow(g, (p - 1) // order, p)
    assert pow(w, order, p) == 1 and all(pow(w, order // f, p) != 1 for f in (2, 3) if order % f == 0)
    return w


if any(os.environ.get(_k) for _k in ("SDA_NO_XCD_MAP", "SDA_NO_NARROW", "SDA_NO_LAZY", "SDA_NO_NGEMM")):
    capi.use_test_hooks()                                        # the knob table exists in libsda_hip_test.so only
lib = capi.load()
for _k in ("SDA_NO_XCD_MAP", "SDA_NO_NARROW", "SDA_NO_LAZY", "SDA_NO_NGEMM"):
    if os.environ.get(_k):
        capi.check(lib.sda_debug_set_knob(_k.encode(), 1))
ONLY = os.environ.get("ONLY_SMALL_G")
NARROW_ONLY = os.environ.get("NARROW_ONLY")
dim = 1 << 20
TSS_P1, TSS_P2 = 746497, 5038849          # tss's own primes: p - 1 = 2^10 * 3^6 and 2^8 * 3^9 (the uint32_t transform kernel)
SHAPES = [(100, 155, 728, 500, None), (100, 155, 19682, 40, None), (100, 155, 2186, 200, None), (40, 23, 242, 500, None), (70, 57, 242, 500, None),
          (100, 155, 728, 500, TSS_P1), (100, 155, 19682, 40, TSS_P2), (100, 155, 2186, 200, TSS_P2), (40, 23, 242, 500, TSS_P1),
          (70, 57, 242, 500, TSS_P1)]
for (k, t, n, P, small) in SHAPES:
    if ONLY and n not in (19682, 2186):
        continue
    if NARROW_ONLY and not small:
        continue
    p = small or (P62B if n + 1 == 19683 else P62)
    sch = crypto.PackedShamir(k, n, t, p, root(p, k + t + 1), root(p, n + 1))
    gen = crypto.ShareGenerator(sch)
    sec = DeviceBuffer(P * dim)
    capi.check(lib.sda_fill_synthetic_dev(sec.ptr, P, dim, dim, 0, 3, p, None))
    B = (dim + k - 1) // k
    Bs = (B + 15) // 16 * 16
    out = DeviceBuffer(n * P * Bs)

    def run():
        gen.generate_batch_dev(sec.ptr, P, dim, dim, out.ptr, Bs, P * Bs, first_participant=0)
    run(); synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(); synchronize(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    wr = n * P * B * 8
    print(f"p={p} k={k} t={t} n={n} ({P} participants): {dt*1e3:.2f} ms  {P*dim/dt/1e9:.2f} Gelem/s  shares written {wr/1e9:.1f} GB = {wr/dt/1e12:.2f} TB/s", flush=True)
    del out, sec, gen
