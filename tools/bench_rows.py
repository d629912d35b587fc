#!/usr/bin/env python3
"""Per-row measurements for the SURVEY.md 8(a) kernels that are not on bench.py's headline path:
reconstruct (R0/R1), ChaCha mask expansion/combination (M2), full masking (M1), unmask (K6), the
cross-GPU partial-sum reducer (X1).  Prints one JSON object; run on the GPU box."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, synchronize  # noqa: E402

P62 = 4611686006577364993
W = {8: 631229665360524489, 9: 3451275676410824977}
lib = capi.load()


def timed(fn, reps=5):
    fn()
    synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


out = {}
# R0/R1: packed reconstruct, BASELINE config 5 reveal: dim 16 Mi, k=3, t=1, n'=4 clerk sums resident in HBM
dim = 1 << 24
k, t, n = 3, 1, 8
sch = crypto.PackedShamir(k, n, t, P62, W[8], W[9])
B = (dim + k - 1) // k
rows = 4
sums = DeviceBuffer(rows * B)
capi.check(lib.sda_fill_synthetic_dev(sums.ptr, rows, B, B, 0, 3, P62, None))
outb = DeviceBuffer(dim)
rec = crypto.SecretReconstructor(sch, dim)
dt = timed(lambda: rec.reconstruct_dev([0, 1, 2, 3], sums.ptr, B, B, outb.ptr, dim))
out["packed_reconstruct_dim16Mi_k3_n4"] = {"ms": dt * 1e3, "secrets_per_s": dim / dt,
                                           "GBps_algorithmic": (rows * B * 8 + dim * 8) / dt / 1e9}
del sums, outb

# M2: ChaCha mask combine (rand-0.3 compatible), P seeds x dim 1 Mi  (recipient side, chacha.rs:56-77)
dim = 1 << 20
for P in (256, 4096):
    seeds = np.random.default_rng(1).integers(0, 1 << 32, size=(P, 4), dtype=np.int64)
    mc = crypto.MaskCombiner(crypto.ChaCha(P62, dim, 128))
    dt = timed(lambda: mc.combine(list(seeds)), reps=3)
    out[f"chacha_mask_combine_P{P}_dim1Mi"] = {"ms": dt * 1e3, "masks_per_s": P * dim / dt,
                                               "note": "host call: includes seed upload, flag readback, 8 MB result download"}

# M1: full masking with the device CSPRNG, dim 16 Mi (host buffers: PCIe included)
dim = 1 << 24
secrets = np.arange(dim, dtype=np.int64)
m = crypto.SecretMasker(crypto.Full(P62))
dt = timed(lambda: m.mask(secrets), reps=3)
out["full_mask_host_dim16Mi"] = {"ms": dt * 1e3, "elements_per_s": dim / dt, "note": "host buffers, PCIe-bound"}

# X1: cross-GPU reducer: 8 parts of the config-5 partial sums [8][5592406]
parts, L = 8, 8 * 5592406
d = DeviceBuffer(parts * L)
capi.check(lib.sda_fill_synthetic_dev(d.ptr, parts, L, L, 0, 5, P62, None))
o = DeviceBuffer(L)
dt = timed(lambda: capi.check(lib.sda_modsum_parts_dev(P62, d.ptr, parts, L, L, o.ptr, None)))
out["modsum_parts_8x358MB"] = {"ms": dt * 1e3, "GBps_algorithmic": (parts + 1) * L * 8 / dt / 1e9}
del d, o

# 8f rank 1: wire codec on one clerk job tile: 2000 participants x L = 349526 canonical 62-bit shares
rows, L, stride = 2000, 349526, 349536
vals = DeviceBuffer(rows * stride)
capi.check(lib.sda_fill_synthetic_dev(vals.ptr, rows, stride, stride, 0, 9, P62, None))
codec = crypto.VarintCodec()
cap = rows * L * 10
d_bytes = DeviceBuffer((cap + 7) // 8)
d_off = DeviceBuffer(rows + 1)
total = [0]
def enc():
    total[0] = codec.encode_dev(vals.ptr, rows, L, stride, d_bytes.ptr, cap, d_off.ptr)
dt = timed(enc, reps=3)
nv = rows * L
out["varint_encode_2000x349526"] = {"ms": dt * 1e3, "values_per_s": nv / dt, "wire_bytes": total[0],
                                    "GBps_algorithmic": (nv * 8 + total[0]) / dt / 1e9}
dec = DeviceBuffer(rows * stride)
st = DeviceBuffer(1).zero()
dt = timed(lambda: codec.decode_dev(d_bytes.ptr, total[0], d_off.ptr, rows, L, dec.ptr, stride, st.ptr), reps=3)
assert st.to_numpy()[0] == 0
out["varint_decode_2000x349526"] = {"ms": dt * 1e3, "values_per_s": nv / dt,
                                    "GBps_algorithmic": (nv * 8 + total[0]) / dt / 1e9}
comb = crypto.ShareCombiner(crypto.Additive(3, P62))
o2 = DeviceBuffer(L)
def dec_comb():
    codec.decode_dev(d_bytes.ptr, total[0], d_off.ptr, rows, L, dec.ptr, stride, st.ptr)
    comb.begin_dev(1, L); comb.update_dev(dec.ptr, 0, rows, stride); comb.finish_dev(o2.ptr)
dt = timed(dec_comb, reps=3)
out["varint_decode_then_clerk_sum_2000x349526"] = {"ms": dt * 1e3, "values_per_s": nv / dt}
print(json.dumps(out, indent=1))
