#!/usr/bin/env python3
"""BASELINE config 2 as a clerk and a recipient see it: additive 3-way sharing + FULL masking over the 62-bit modulus, dim
1 Mi, with everything the reference puts on the wire, device resident.  Per tile of P participants:

  participant side (participate.rs:52-113)   full mask (mask, masked = secrets + mask; :53-54) -> additive share-gen of the
                                             MASKED secrets (:75-76) -> zig-zag varint encode of the 3 P share vectors and of
                                             the P mask vectors -> one sealed box per (clerk, participant) + one per
                                             participant for the recipient (the mask, :56-72)
  server (snapshot.rs:4-47)                  nothing to do: boxes are written job-major
  clerk side (clerk.rs:63-107), per clerk    open the P boxes -> clerk sums straight from the varint bytes
  recipient (receive.rs:101-116)             open the P mask boxes -> mask combine (column sum mod q, full.rs:37-52)
and at the end (receive.rs:140-156) reconstruct (sum of the 3 clerk sums) -> unmask -> positive, checked against the sum of
the secrets.  Prints one JSON object; run on the GPU box.  TILE / TILES override the job size."""
import ctypes as C
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, DeviceBytes, synchronize  # noqa: E402

P62 = 4611686006577364993
lib = capi.load()
n, dim = 3, 1 << 20
P, tiles = int(os.environ.get("TILE", "512")), int(os.environ.get("TILES", "4"))   # 3 x 512 = 1536 share rows: the streaming clerk path
sch, msch = crypto.Additive(n, P62), crypto.Full(P62)
parties = n + 1                                                 # three clerks and the recipient
rows = parties * P
secrets = DeviceBuffer(P * dim)
capi.check(lib.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 0x5DA5DA5DA5DA5DA5, P62, None))
# [clerk 0 shares | clerk 1 | clerk 2 | masks] x P rows of dim values: the masks sit behind the shares so that ONE encode and
# ONE seal call serve all four recipients
values = DeviceBuffer(rows * dim)
masked = DeviceBuffer(P * dim)
d_shares, d_masks = values.ptr, values.ptr + n * P * dim * 8
codec, box = crypto.VarintCodec(), crypto.SealedBox()
vslot = codec.slot_size(dim)
bslot = vslot + 48
wire, wlen = DeviceBytes(rows * vslot), DeviceBytes(rows * 8)
boxes, blen = DeviceBytes(rows * bslot), DeviceBytes(rows * 8)
plain, plen = DeviceBytes(rows * vslot), DeviceBytes(rows * 8)
status = DeviceBytes(4).zero()
masker = crypto.SecretMasker(msch)
gen = crypto.ShareGenerator(sch)
comb = crypto.ShareCombiner(sch)                                # the three clerks' sums
mcomb = crypto.ShareCombiner(crypto.Additive(2, P62))           # the recipient's mask combine: the same column sum (full.rs:37-52)
sks = [bytes([c + 1]) * 32 for c in range(parties)]
pks = [box.public_key(sk) for sk in sks]


def ev():
    e = C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(e)))
    return e


stage_ms = {"full_mask": 0.0, "share_gen": 0.0, "varint_encode": 0.0, "seal": 0.0, "open": 0.0, "decode_and_sums": 0.0}


def stage(name, fn):
    a, b = ev(), ev()
    capi.check(lib.sda_event_record(a, None))
    fn()
    capi.check(lib.sda_event_record(b, None))
    synchronize()
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(a, b, C.byref(ms)))
    stage_ms[name] += ms.value


hip = C.CDLL("libamdhip64.so")
party_boxes = [crypto.SealedBox() for _ in range(parties)]      # every party is its own handle and HIP stream
party_streams = []
for _ in range(parties):
    sp = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(sp)) == 0
    party_streams.append(sp)


def open_all():
    synchronize()
    for c in range(parties):
        party_boxes[c].open_rows_dev(pks[c], sks[c], boxes.ptr + c * P * bslot, bslot, blen.ptr + c * P * 8, P, bslot,
                                     plain.ptr + c * P * vslot, vslot, plen.ptr + c * P * 8, status.ptr,
                                     stream=party_streams[c].value)
    for sp in party_streams:
        assert hip.hipStreamSynchronize(sp) == 0


def sums():
    comb.update_encoded_rows_dev(codec, plain.ptr, vslot, plen.ptr, n * P, status.ptr)
    mcomb.update_encoded_rows_dev(codec, plain.ptr + n * P * vslot, vslot, plen.ptr + n * P * 8, P, status.ptr)


def tile(i, timed):
    run = stage if timed else (lambda _n, fn: fn())
    run("full_mask", lambda: masker.mask_batch_dev(secrets.ptr, P, dim, dim, d_masks, dim, masked.ptr, dim, first_participant=i * P))
    run("share_gen", lambda: gen.generate_batch_dev(masked.ptr, P, dim, dim, d_shares, dim, P * dim, first_participant=i * P))
    run("varint_encode", lambda: codec.encode_rows_dev(values.ptr, rows, dim, dim, wire.ptr, vslot, wlen.ptr))
    run("seal", lambda: box.seal_rows_dev(pks, P, wire.ptr, vslot, wlen.ptr, rows, vslot, boxes.ptr, bslot, blen.ptr))
    run("open", open_all)
    run("decode_and_sums", sums)


comb.begin_dev(n, dim); mcomb.begin_dev(1, dim)
tile(0, False)                                                  # warm-up (allocations), its sums are discarded
synchronize()
comb.begin_dev(n, dim); mcomb.begin_dev(1, dim)
t0, t1 = ev(), ev()
capi.check(lib.sda_event_record(t0, None))
for i in range(tiles):
    tile(i, True)
capi.check(lib.sda_event_record(t1, None))
synchronize()
ms = C.c_float()
capi.check(lib.sda_event_elapsed_ms(t0, t1, C.byref(ms)))
box_bytes = int(np.frombuffer(blen.to_bytes(), dtype="<u8").sum())

# recipient (receive.rs:140-156): reconstruct the masked sum from the three clerk sums, unmask with the combined mask
csums, msum = DeviceBuffer(n * dim), DeviceBuffer(dim)
comb.finish_dev(csums.ptr); mcomb.finish_dev(msum.ptr)
rec = crypto.SecretReconstructor(sch, dim)
masked_sum, out = DeviceBuffer(dim), DeviceBuffer(dim)
rec.reconstruct_dev([0, 1, 2], csums.ptr, dim, dim, masked_sum.ptr, dim)
crypto.SecretUnmasker(msch).unmask_dev(msum.ptr, masked_sum.ptr, dim, out.ptr)
col = crypto.ShareCombiner(crypto.Additive(3, P62))
col.begin_dev(1, dim)
col.update_dev(secrets.ptr, 0, P, dim)
colsum = DeviceBuffer(dim)
col.finish_dev(colsum.ptr)
want = (colsum.to_numpy().astype(object) * tiles) % P62
got = crypto.positive(out.to_numpy(), P62) if hasattr(crypto, "positive") else out.to_numpy()
ok = bool(np.array_equal(np.asarray(got).astype(object), want)) and status.to_bytes() == bytes(4)
elements = tiles * P * dim
print(json.dumps({
    "job": f"{tiles} tiles x {P} participants x dim {dim}, additive {n}-way + full mask, 62-bit modulus; {rows} sealed boxes per tile "
           f"({n} clerks + the recipient's mask)",
    "box_bytes_per_tile": box_bytes, "box_bytes_per_secret": box_bytes / (P * dim),
    "ms_per_tile": ms.value / tiles, "elements_per_s": elements / (ms.value * 1e-3),
    "stage_ms_per_tile": {s: v / tiles for s, v in stage_ms.items()},
    "whole_config2_job_s": 10_000 * dim / (elements / (ms.value * 1e-3)),
    "verified_reveal_equals_sum_of_secrets": ok}, indent=1))
