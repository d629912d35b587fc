#!/usr/bin/env python3
"""Share-gen rate of several (k, t) splits, compiled shapes vs the generic fallback; run on the GPU box."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sda_amd import capi, crypto
from sda_amd.device import DeviceBuffer, synchronize
P62 = 4611686006577364993
W = {8: 631229665360524489, 9: 3451275676410824977, 16: None, 27: None}
lib = capi.load()
dim, P = 1 << 20, 500
sec = DeviceBuffer(P * dim)
capi.check(lib.sda_fill_synthetic_dev(sec.ptr, P, dim, dim, 0, 3, P62, None))
W[16], W[27], W[32], W[64], W[3] = 2589100645267092065, 365137883145458390, 1942624553499164220, 2724396144719537715, 3
for (k, t, n, o2, o3) in [(4, 3, 8, 8, 9), (5, 2, 8, 8, 9), (6, 1, 8, 8, 9), (7, 0, 8, 8, 9), (3, 4, 8, 8, 9), (8, 7, 26, 16, 27),
                          (6, 2, 8, 16, 9), (9, 6, 26, 16, 27),    # run-time (k, t) kernel
                          (3, 4, 80, 8, 3), (8, 7, 80, 16, 3), (8, 7, 242, 16, 3), (10, 7, 26, 32, 27), (20, 11, 40, 32, 3),   # the same, matrix in global memory
                          (12, 3, 26, 16, 27), (10, 5, 26, 16, 27), (8, 2, 26, 16, 27), (3, 1, 8, 8, 9),    # compiled, 5-term groups where they help
                          (20, 13, 80, 64, 3)]:                    # generic kernel (k + t > 32, not a tss shape)
    sch = crypto.PackedShamir(k, n, t, P62, W[o2], W[o3])
    gen = crypto.ShareGenerator(sch)
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
    print(f"k={k} t={t} n={n}: {dt*1e3:.2f} ms  {P*dim/dt/1e9:.1f} Gelem/s")
