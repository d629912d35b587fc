#!/usr/bin/env python3
"""Per-row measurements for the SURVEY.md 8(a) kernels that are not on bench.py's headline path:
reconstruct (R0/R1), ChaCha mask expansion/combination (M2), full masking (M1), unmask (K6), the
cross-GPU partial-sum reducer (X1).  The wire-format rows are in tools/bench_wire.py.  Prints one JSON object; run on
the GPU box."""
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
    dt = timed(lambda: mc.combine(seeds), reps=3)
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

# M1 on device: full masking of a resident tile (participate.rs:52-54 x 2000 participants), masks from the device CSPRNG
P, dim = 2000, 1 << 20
sec = DeviceBuffer(P * dim)
capi.check(lib.sda_fill_synthetic_dev(sec.ptr, P, dim, dim, 0, 11, P62, None))
d_mask, d_masked = DeviceBuffer(P * dim), DeviceBuffer(P * dim)
mk = crypto.SecretMasker(crypto.Full(P62))
dt = timed(lambda: mk.mask_batch_dev(sec.ptr, P, dim, dim, d_mask.ptr, dim, d_masked.ptr, dim), reps=3)
out["full_mask_batch_dev_2000x1Mi"] = {"ms": dt * 1e3, "elements_per_s": P * dim / dt,
                                      "GBps_algorithmic": P * dim * 24 / dt / 1e9,
                                      "note": "8 B read + 8 B mask + 8 B masked per element, one ChaCha20 draw per element"}
# recipient: combine the 2000 mask vectors (full.rs:37-52 == the clerk-sum kernel), then unmask (full.rs:54-67)
cm = crypto.ShareCombiner(crypto.Additive(3, P62))
tot, o3 = DeviceBuffer(dim), DeviceBuffer(dim)
def comb_masks():
    cm.begin_dev(1, dim); cm.update_dev(d_mask.ptr, 0, P, dim); cm.finish_dev(tot.ptr)
dt = timed(comb_masks, reps=3)
out["full_mask_combine_dev_2000x1Mi"] = {"ms": dt * 1e3, "elements_per_s": P * dim / dt, "GBps_algorithmic": P * dim * 8 / dt / 1e9}
um = crypto.SecretUnmasker(crypto.Full(P62))
dt = timed(lambda: um.unmask_dev(tot.ptr, d_masked.ptr, dim, o3.ptr), reps=5)
out["unmask_dev_1Mi"] = {"ms": dt * 1e3, "elements_per_s": dim / dt}
# M2 on device: ChaCha-seeded masking of the same tile (chacha.rs:24-54 x 2000 participants; seeds from OS entropy)
mkc = crypto.SecretMasker(crypto.ChaCha(P62, dim, 128))
d_seedw = DeviceBuffer(P * 4)
dt = timed(lambda: mkc.mask_batch_dev(sec.ptr, P, dim, dim, d_seedw.ptr, 4, d_masked.ptr, dim), reps=3)
out["chacha_mask_batch_dev_2000x1Mi"] = {"ms": dt * 1e3, "elements_per_s": P * dim / dt,
                                        "note": "rand-0.3 ChaChaRng expansion of one seed per participant, 8 B read + 8 B written per element"}
del sec, d_mask, d_masked
print(json.dumps(out, indent=1))
