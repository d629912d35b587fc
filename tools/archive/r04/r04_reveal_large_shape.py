#!/usr/bin/env python3
"""reveal (Lagrange reconstruction from t + k clerk rows) of a LARGE shape: tss's PSS_155_728_100, 1 Mi secrets from 255 rows,
over tss's prime and over the 62-bit prime - the any-shape kernel (packed_reconstruct_kernel), checked against the oracle on a prefix"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto
from sda_amd.device import DeviceBuffer, synchronize
from oracle import coracle
lib = capi.load()
out = {}
k, t, n = 100, 155, 728
for name, p, w2, w3 in (("746497", 746497, 95660, 610121), ("p62", 4611686006577364993, None, None)):
    if w2 is None:
        g = next(g for g in range(2, 500) if all(pow(g, (p - 1) // f, p) != 1 for f in (2, 3)))
        w2, w3 = pow(g, (p - 1) // 256, p), pow(g, (p - 1) // 729, p)
    dim = 1 << 20
    B = (dim + k - 1) // k
    rows = k + t
    idx = sorted(np.random.default_rng(1).choice(n, size=rows, replace=False).tolist())
    d_sh = DeviceBuffer(rows * B)
    capi.check(lib.sda_fill_synthetic_dev(d_sh.ptr, rows, B, B, 0, 5, p, None))
    d_out = DeviceBuffer(dim)
    rec = crypto.SecretReconstructor(crypto.PackedShamir(k, n, t, p, w2, w3), dim)
    rec.reconstruct_dev(idx, d_sh.ptr, B, B, d_out.ptr, dim)
    synchronize()
    e0, e1 = C.c_void_p(), C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(e0))); capi.check(lib.sda_event_create(C.byref(e1)))
    capi.check(lib.sda_event_record(e0, None))
    for _ in range(5):
        rec.reconstruct_dev(idx, d_sh.ptr, B, B, d_out.ptr, dim)
    capi.check(lib.sda_event_record(e1, None))
    synchronize()
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(e0, e1, C.byref(ms)))
    sh = d_sh.to_numpy().reshape(rows, B)
    nb = 40                                                        # the oracle on the first 40 batches
    want = coracle.packed_reconstruct(p, k, t, w2, w3, nb * k, idx, np.ascontiguousarray(sh[:, :nb]))
    out[name] = {"ms": ms.value / 5, "secrets": dim, "rows": rows, "matches_oracle_prefix": bool(np.array_equal(d_out.to_numpy()[:nb * k], want))}
print(json.dumps(out, indent=1))
