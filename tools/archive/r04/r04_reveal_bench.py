#!/usr/bin/env python3
"""reveal (Lagrange reconstruction) over a narrow prime: the one-limb kernel against the 64-bit kernel, 16 Mi secrets"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto
from sda_amd.device import DeviceBuffer, synchronize
lib = capi.load()
P31 = 2147482801
out = {}
for name, (k, t, n, w2, w3) in {"(3,1,8)": (3, 1, 8, 495332030, 1761729792), "(8,7,26)": (8, 7, 26, 1541819067, 638656353)}.items():
    dim = 1 << 24
    B = (dim + k - 1) // k
    rows = k + t
    d_sh = DeviceBuffer(rows * B)
    capi.check(lib.sda_fill_synthetic_dev(d_sh.ptr, rows, B, B, 0, 5, P31, None))
    d_out = DeviceBuffer(dim)
    for label, knob in (("narrow", 0), ("wide", 1)):
        capi.check(lib.sda_debug_set_knob(b"SDA_NO_NARROW", knob))
        rec = crypto.SecretReconstructor(crypto.PackedShamir(k, n, t, P31, w2, w3), dim)
        idx = list(range(rows))
        rec.reconstruct_dev(idx, d_sh.ptr, B, B, d_out.ptr, dim)
        synchronize()
        e0, e1 = C.c_void_p(), C.c_void_p()
        capi.check(lib.sda_event_create(C.byref(e0))); capi.check(lib.sda_event_create(C.byref(e1)))
        capi.check(lib.sda_event_record(e0, None))
        for _ in range(10):
            rec.reconstruct_dev(idx, d_sh.ptr, B, B, d_out.ptr, dim)
        capi.check(lib.sda_event_record(e1, None))
        synchronize()
        ms = C.c_float()
        capi.check(lib.sda_event_elapsed_ms(e0, e1, C.byref(ms)))
        out[f"{name} {label}"] = {"ms": ms.value / 10, "GBps": 8 * (rows * B + dim) / (ms.value / 10 * 1e-3) / 1e9,
                                  "checksum": int(d_out.to_numpy()[:4096].astype(object).sum())}
print(json.dumps(out, indent=1))
