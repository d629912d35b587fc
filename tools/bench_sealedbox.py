#!/usr/bin/env python3
"""A clerk's whole job with sealed payloads, device resident: P participants' share vectors for
ONE clerk (config-3 shape: B = 349526 shares each, ~3.1 MB of varint bytes) are encoded and sealed on the device
(participant side, participate.rs:82-101), then - the timed clerk side, clerk.rs:78-86 - opened, decoded and summed:
    sda_sealedbox_open_rows_dev -> sda_share_combiner_update_varint_rows_dev -> finish
Prints one JSON object; run on the GPU box.  ROWS / VALUES / REPS override the job size."""
import ctypes as C
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, DeviceBytes, synchronize  # noqa: E402

P62 = 4611686006577364993
lib = capi.load()
P, L, reps = int(os.environ.get("ROWS", "2000")), int(os.environ.get("VALUES", "349526")), int(os.environ.get("REPS", "5"))
shares = DeviceBuffer(P * L)
capi.check(lib.sda_fill_synthetic_dev(shares.ptr, P, L, L, 0, 0x5DA5DA5DA5DA5DA5, P62, None))
codec, box = crypto.VarintCodec(), crypto.SealedBox()
vslot = codec.slot_size(L)
wire, wlen = DeviceBytes(P * vslot), DeviceBytes(P * 8)
bslot = vslot + 48
boxes, blen = DeviceBytes(P * bslot), DeviceBytes(P * 8)
plain, plen = DeviceBytes(P * vslot), DeviceBytes(P * 8)
status = DeviceBytes(4).zero()
sk = bytes(range(1, 33))
pk = (C.c_uint8 * 32)()
# the clerk's public key = X25519(sk, 9): computed by sealing an empty message to the base point?  No - simplest: the
# library has no key-generation entry point (the keystore is out of scope); derive it with the oracle-free trick that
# a sealed box's first 32 bytes are X25519(esk, 9)
pk = box.public_key(sk)


def ev():
    e = C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(e)))
    return e


def timed(fn, n):
    fn(); synchronize()
    a, b = ev(), ev()
    capi.check(lib.sda_event_record(a, None))
    for _ in range(n):
        fn()
    capi.check(lib.sda_event_record(b, None))
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(a, b, C.byref(ms)))
    return ms.value / n


enc_ms = timed(lambda: codec.encode_rows_dev(shares.ptr, P, L, L, wire.ptr, vslot, wlen.ptr), reps)
seal_ms = timed(lambda: box.seal_rows_dev([pk], P, wire.ptr, vslot, wlen.ptr, P, vslot, boxes.ptr, bslot, blen.ptr), reps)
box_bytes = int(np.frombuffer(blen.to_bytes(), dtype="<u8").sum())
comb = crypto.ShareCombiner(crypto.Additive(3, P62))
open_ms = timed(lambda: box.open_rows_dev(pk, sk, boxes.ptr, bslot, blen.ptr, P, bslot, plain.ptr, vslot, plen.ptr, status.ptr), reps)


def clerk():
    comb.begin_dev(1, L)
    box.open_rows_dev(pk, sk, boxes.ptr, bslot, blen.ptr, P, bslot, plain.ptr, vslot, plen.ptr, status.ptr)
    comb.update_encoded_rows_dev(codec, plain.ptr, vslot, plen.ptr, P, status.ptr)


job_ms = timed(clerk, reps)
sums = DeviceBuffer(L)
comb.finish_dev(sums.ptr)
# verify against the direct clerk sum of the plaintext shares
direct = crypto.ShareCombiner(crypto.Additive(3, P62))
direct.begin_dev(1, L)
direct.update_dev(shares.ptr, 0, P, L)
want = DeviceBuffer(L)
direct.finish_dev(want.ptr)
ok = bool(np.array_equal(sums.to_numpy(), want.to_numpy())) and status.to_bytes() == bytes(4)
print(json.dumps({"job": f"{P} sealed share vectors x {L} values (config-3 clerk job tile), 62-bit residues",
                  "box_bytes_total": box_bytes, "box_bytes_per_value": box_bytes / (P * L),
                  "participant_side": {"varint_encode_rows_ms": enc_ms, "seal_rows_ms": seal_ms,
                                       "seal_GBps": box_bytes / (seal_ms * 1e-3) / 1e9},
                  "clerk_side": {"open_rows_ms": open_ms, "open_GBps": box_bytes / (open_ms * 1e-3) / 1e9,
                                 "open_decode_sum_ms": job_ms, "values_per_s": P * L / (job_ms * 1e-3),
                                 "box_GBps": box_bytes / (job_ms * 1e-3) / 1e9},
                  "verified_sums_equal_plaintext_clerk_sum": ok}, indent=1))
