#!/usr/bin/env python3
"""The whole simulated aggregation over the wire format, device resident: per tile of P participants
share-gen -> slotted varint encode of the n x P share vectors -> streaming clerk sums straight from the wire bytes
(no decoded tile) -> (at the end) reconstruct == sum of secrets.  Prints one JSON object; run on the GPU box."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, synchronize  # noqa: E402

P62 = 4611686006577364993
W8, W9 = 631229665360524489, 3451275676410824977
lib = capi.load()
k, t, n, dim = 3, 1, 8, 1 << 20
P, tiles = int(os.environ.get("TILE", "2000")), int(os.environ.get("TILES", "5"))
sch = crypto.PackedShamir(k, n, t, P62, W8, W9)
B = (dim + k - 1) // k
Bs = (B + 15) // 16 * 16
secrets = DeviceBuffer(P * dim)
capi.check(lib.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 0x5DA5DA5DA5DA5DA5, P62, None))
shares = DeviceBuffer(n * P * Bs)
codec = crypto.VarintCodec()
slot = codec.slot_size(B)
wire = DeviceBuffer(n * P * slot // 8 + 2)
lens = DeviceBuffer(n * P)
st = DeviceBuffer(1).zero()
gen = crypto.ShareGenerator(sch)
comb = crypto.ShareCombiner(sch)


def tile(i):
    gen.generate_batch_dev(secrets.ptr, P, dim, dim, shares.ptr, Bs, P * Bs, first_participant=i * P)
    codec.encode_rows_dev(shares.ptr, n * P, B, Bs, wire.ptr, slot, lens.ptr)
    comb.update_encoded_rows_dev(codec, wire.ptr, slot, lens.ptr, n * P, st.ptr)


comb.begin_dev(n, B)
tile(0)
synchronize()
comb.begin_dev(n, B)
t0 = time.perf_counter()
for i in range(tiles):
    tile(i)
sums = DeviceBuffer(n * B)
comb.finish_dev(sums.ptr)
synchronize()
dt = time.perf_counter() - t0
assert st.to_numpy()[0] == 0
wire_bytes = int(lens.to_numpy().sum())
# verify: reconstruct(clerk sums) == tiles x column sums of the secrets tile
rec = crypto.SecretReconstructor(sch, dim)
out = DeviceBuffer(dim)
rec.reconstruct_dev([0, 1, 2, 3], sums.ptr, B, B, out.ptr, dim)
cs = crypto.ShareCombiner(crypto.Additive(2, P62))
cs.begin_dev(1, dim)
for _ in range(tiles):
    cs.update_dev(secrets.ptr, 0, P, dim)
exp = DeviceBuffer(dim)
cs.finish_dev(exp.ptr)
ok = bool(np.array_equal(out.to_numpy(), exp.to_numpy()))
print(json.dumps({"pipeline": "share-gen -> slotted varint encode -> clerk sums from the wire bytes", "shape": "k=3 t=1 n=8 dim 1 Mi",
                  "tile_participants": P, "tiles": tiles, "ms_per_tile": dt / tiles * 1e3, "elements_per_s": tiles * P * dim / dt,
                  "wire_bytes_per_tile": wire_bytes, "wire_bytes_per_share": wire_bytes / (n * P * B),
                  "verified_reconstruct_equals_sum": ok}, indent=1))
