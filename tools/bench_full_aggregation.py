#!/usr/bin/env python3
"""A whole aggregation with everything the reference puts on the wire, device resident (config-3 shape: packed Shamir
k=3, t=1, n=8 over the 62-bit prime, dim 1 Mi).  Per tile of P participants:

  participant side (participate.rs:52-113)   share-gen -> zig-zag varint encode of the n x P share vectors -> one sealed
                                             box per (clerk, participant), sealed to that clerk's public key
  server (snapshot.rs:4-47)                  nothing to do: the boxes are written clerk-major, clerk c's job is rows c*P..
  clerk side (clerk.rs:63-107), per clerk    open the P boxes with the clerk's key -> clerk sums straight from the varint
                                             bytes (no decoded tile)
and at the end (receive.rs:80-157) finish -> reveal from t + k clerks, checked against the sum of the secrets.
Prints one JSON object; run on the GPU box.  TILE / TILES override the job size."""
import ctypes as C
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, DeviceBytes, synchronize  # noqa: E402

P62 = 4611686006577364993
W8, W9 = 631229665360524489, 3451275676410824977
if any(os.environ.get(_k) for _k in ("SDA_WIRE_WG_PER_CU", "SDA_SBOX_WG_PER_CU")):
    capi.use_test_hooks()                                        # the knob table exists in libsda_hip_test.so only
lib = capi.load()
for _knob in ("SDA_WIRE_WG_PER_CU", "SDA_SBOX_WG_PER_CU"):       # residency caps for the two-stream schedule (include/sda_hip_debug.h)
    if os.environ.get(_knob):
        capi.check(lib.sda_debug_set_knob(_knob.encode(), int(os.environ[_knob])))
k, t, n, dim = 3, 1, 8, 1 << 20
P, tiles = int(os.environ.get("TILE", "1000")), int(os.environ.get("TILES", "4"))
# PRIME=<p>: the same job over another prime with roots of order 8 and 9 - e.g. tss's own 746497 or the 31-bit 2147482801: the
# reference's valid domain (tss multiplies i64 residues without widening), where a share is a 3- to 5-byte varint instead of 9
if os.environ.get("PRIME"):
    P62 = int(os.environ["PRIME"])
    assert (P62 - 1) % 72 == 0
    _g = next(g for g in range(2, 500) if all(pow(g, (P62 - 1) // f, P62) != 1 for f in (2, 3)))
    W8, W9 = pow(_g, (P62 - 1) // 8, P62), pow(_g, (P62 - 1) // 9, P62)
sch = crypto.PackedShamir(k, n, t, P62, W8, W9)
B = (dim + k - 1) // k
Bs = (B + 15) // 16 * 16
rows = n * P
secrets = DeviceBuffer(P * dim)
capi.check(lib.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 0x5DA5DA5DA5DA5DA5, P62, None))
shares = DeviceBuffer(rows * Bs)
codec, box = crypto.VarintCodec(), crypto.SealedBox()
vslot = codec.slot_size(B)
bslot = vslot + 48
wire, wlen = DeviceBytes(rows * vslot), DeviceBytes(rows * 8)
boxes, blen = DeviceBytes(rows * bslot), DeviceBytes(rows * 8)
plain, plen = DeviceBytes(rows * vslot), DeviceBytes(rows * 8)
status = DeviceBytes(4).zero()
gen = crypto.ShareGenerator(sch)
comb = crypto.ShareCombiner(sch)
sks = [bytes([c + 1]) * 32 for c in range(n)]
pks = [box.public_key(sk) for sk in sks]                        # X25519(sk, 9): the clerks' public keys


def ev():
    e = C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(e)))
    return e


stage_ms = {"share_gen": 0.0, "varint_encode": 0.0, "seal": 0.0, "open": 0.0, "decode_and_clerk_sum": 0.0}


def stage(name, fn):
    a, b = ev(), ev()
    capi.check(lib.sda_event_record(a, None))
    fn()
    capi.check(lib.sda_event_record(b, None))
    synchronize()
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(a, b, C.byref(ms)))
    stage_ms[name] += ms.value


# every clerk is its own party: own handle (scratch) and own HIP stream, so the clerks' short latency-bound setup kernels
# overlap instead of queueing behind each other (the ABI takes any hipStream_t; the tool makes them with the HIP runtime)
hip = C.CDLL("libamdhip64.so")
clerk_boxes = [crypto.SealedBox() for _ in range(n)]
clerk_streams = []
for _ in range(n):
    sp = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(sp)) == 0
    clerk_streams.append(sp)


def open_all():
    synchronize()                                               # the boxes were sealed on the default stream
    for c in range(n):
        clerk_boxes[c].open_rows_dev(pks[c], sks[c], boxes.ptr + c * P * bslot, bslot, blen.ptr + c * P * 8, P, bslot,
                                     plain.ptr + c * P * vslot, vslot, plen.ptr + c * P * 8, status.ptr,
                                     stream=clerk_streams[c].value)
    for sp in clerk_streams:
        assert hip.hipStreamSynchronize(sp) == 0


def tile(i, timed):
    run = stage if timed else (lambda _n, fn: fn())
    run("share_gen", lambda: gen.generate_batch_dev(secrets.ptr, P, dim, dim, shares.ptr, Bs, P * Bs, first_participant=i * P))
    run("varint_encode", lambda: codec.encode_rows_dev(shares.ptr, rows, B, Bs, wire.ptr, vslot, wlen.ptr))
    run("seal", lambda: box.seal_rows_dev(pks, P, wire.ptr, vslot, wlen.ptr, rows, vslot, boxes.ptr, bslot, blen.ptr))
    run("open", open_all)
    run("decode_and_clerk_sum", lambda: comb.update_encoded_rows_dev(codec, plain.ptr, vslot, plen.ptr, rows, status.ptr))


# ---- PIPELINE=1: the same stages software-pipelined over tiles on two HIP streams --------------------------------------------
# The sealed-box kernels are bound by the vector ALUs (Salsa20/20: ~960 instructions per 64 bytes, its instruction floor),
# share generation / varint encode / decode-and-sum by HBM; run back to back each leaves the other resource idle.  Stream A
# carries seal(i) and open(i), stream B (lower priority) decode+sum(i-1), share-gen(i+1) and encode(i+1); wire and plaintext
# buffers are double-buffered, events order producer and consumer.  Boxes, sums and the reveal are the same bytes as in the
# serial schedule (verified below); only what runs beside what changes.
PIPELINE = os.environ.get("PIPELINE", "0") == "1"
if PIPELINE:
    least, greatest = C.c_int(), C.c_int()
    assert hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) == 0
    sA, sB = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreateWithPriority(C.byref(sA), 1, greatest.value) == 0      # hipStreamNonBlocking = 1
    assert hip.hipStreamCreateWithPriority(C.byref(sB), 1, least.value) == 0
    A, Bq = sA.value, sB.value

    def hev():
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0                       # hipEventDisableTiming
        return e

    def record(e, st):
        assert hip.hipEventRecord(e, C.c_void_p(st)) == 0

    def wait(st, e):
        assert hip.hipStreamWaitEvent(C.c_void_p(st), e, 0) == 0

    wire2 = [wire, DeviceBytes(rows * vslot)]
    wlen2 = [wlen, DeviceBytes(rows * 8)]
    plain2 = [plain, DeviceBytes(rows * vslot)]
    plen2 = [plen, DeviceBytes(rows * 8)]
    wire_ready, wire_free = [hev(), hev()], [hev(), hev()]
    plain_ready, plain_free = [hev(), hev()], [hev(), hev()]
    sealed, opened = hev(), [hev() for _ in range(n)]

    def produce(i):                                             # stream B: share-gen + encode of tile i
        j = i % 2
        wait(Bq, wire_free[j])
        gen.generate_batch_dev(secrets.ptr, P, dim, dim, shares.ptr, Bs, P * Bs, first_participant=i * P, stream=Bq)
        codec.encode_rows_dev(shares.ptr, rows, B, Bs, wire2[j].ptr, vslot, wlen2[j].ptr, stream=Bq)
        record(wire_ready[j], Bq)

    def consume(i):                                             # stream B: clerk sums of tile i straight from its plaintext bytes
        j = i % 2
        wait(Bq, plain_ready[j])
        comb.update_encoded_rows_dev(codec, plain2[j].ptr, vslot, plen2[j].ptr, rows, status.ptr, stream=Bq)
        record(plain_free[j], Bq)

    def crypto_stage(i):                                        # stream A (+ the clerks' own streams): seal, then every clerk opens
        j = i % 2
        wait(A, wire_ready[j])
        box.seal_rows_dev(pks, P, wire2[j].ptr, vslot, wlen2[j].ptr, rows, vslot, boxes.ptr, bslot, blen.ptr, stream=A)
        record(wire_free[j], A)
        wait(A, plain_free[j])
        record(sealed, A)
        for c in range(n):
            st = clerk_streams[c].value
            wait(st, sealed)
            clerk_boxes[c].open_rows_dev(pks[c], sks[c], boxes.ptr + c * P * bslot, bslot, blen.ptr + c * P * 8, P, bslot,
                                         plain2[j].ptr + c * P * vslot, vslot, plen2[j].ptr + c * P * 8, status.ptr, stream=st)
            record(opened[c], st)
            wait(A, opened[c])
        record(plain_ready[j], A)

    def run_pipeline(count):
        for e in wire_free + plain_free:
            record(e, A)
        comb.begin_dev(n, B, stream=Bq)
        produce(0)
        for s in range(count + 1):
            if s >= 1:
                consume(s - 1)
            if s + 1 < count:
                produce(s + 1)
            if s < count:
                crypto_stage(s)
        assert hip.hipStreamSynchronize(sA) == 0 and hip.hipStreamSynchronize(sB) == 0
        for sp in clerk_streams:
            assert hip.hipStreamSynchronize(sp) == 0

    import time
    run_pipeline(2)                                             # warm-up (allocations), sums discarded by the begin_dev of the next run
    synchronize()
    t_start = time.perf_counter()
    run_pipeline(tiles)
    synchronize()
    ms_total = (time.perf_counter() - t_start) * 1e3
    box_bytes = int(np.frombuffer(blen.to_bytes(), dtype="<u8").sum())
    sums = DeviceBuffer(n * B)
    comb.finish_dev(sums.ptr)
    rec = crypto.SecretReconstructor(sch, dim)
    out = DeviceBuffer(dim)
    idx = [7, 0, 3, 5]
    picked = DeviceBuffer.from_numpy(np.stack([sums.to_numpy(B, c * B) for c in idx]))
    rec.reconstruct_dev(idx, picked.ptr, B, B, out.ptr, dim)
    col = crypto.ShareCombiner(crypto.Additive(3, P62))
    col.begin_dev(1, dim)
    col.update_dev(secrets.ptr, 0, P, dim)
    colsum = DeviceBuffer(dim)
    col.finish_dev(colsum.ptr)
    want = (colsum.to_numpy().astype(object) * tiles) % P62
    ok = bool(np.array_equal(out.to_numpy().astype(object), want)) and status.to_bytes() == bytes(4)
    elements = tiles * P * dim
    print(json.dumps({
        "schedule": "PIPELINE=1: seal(i) + open(i) on a high-priority stream, decode+sum(i-1) / share-gen(i+1) / encode(i+1) on a "
                    "low-priority stream, double-buffered wire and plaintext tiles; wall clock over the whole run",
        "job": f"{tiles} tiles x {P} participants x dim {dim}, packed Shamir k={k} t={t} n={n}, {P62.bit_length()}-bit prime {P62}; {rows} sealed boxes per tile",
        "box_bytes_per_tile": box_bytes, "box_bytes_per_secret": box_bytes / (P * dim),
        "ms_per_tile": ms_total / tiles, "elements_per_s": elements / (ms_total * 1e-3),
        "whole_config3_job_s": 100_000 * dim / (elements / (ms_total * 1e-3)),
        "verified_reveal_equals_sum_of_secrets": ok}, indent=1))
    sys.exit(0)

comb.begin_dev(n, B)
tile(0, False)                                                  # warm-up (allocations), its sums are discarded
synchronize()
comb.begin_dev(n, B)
t0, t1 = ev(), ev()
capi.check(lib.sda_event_record(t0, None))
for i in range(tiles):
    tile(i, True)
capi.check(lib.sda_event_record(t1, None))
synchronize()
ms = C.c_float()
capi.check(lib.sda_event_elapsed_ms(t0, t1, C.byref(ms)))
box_bytes = int(np.frombuffer(blen.to_bytes(), dtype="<u8").sum())

sums = DeviceBuffer(n * B)
comb.finish_dev(sums.ptr)
rec = crypto.SecretReconstructor(sch, dim)
out = DeviceBuffer(dim)
idx = [7, 0, 3, 5]                                              # any t + k clerks
picked = DeviceBuffer.from_numpy(np.stack([sums.to_numpy(B, c * B) for c in idx]))
rec.reconstruct_dev(idx, picked.ptr, B, B, out.ptr, dim)
# truth: tiles x (column sums of the resident secrets) mod p - every tile re-shares the same secrets with fresh randomness
col = crypto.ShareCombiner(crypto.Additive(3, P62))
col.begin_dev(1, dim)
col.update_dev(secrets.ptr, 0, P, dim)
colsum = DeviceBuffer(dim)
col.finish_dev(colsum.ptr)
want = (colsum.to_numpy().astype(object) * tiles) % P62
ok = bool(np.array_equal(out.to_numpy().astype(object), want)) and status.to_bytes() == bytes(4)
elements = tiles * P * dim
print(json.dumps({
    "job": f"{tiles} tiles x {P} participants x dim {dim}, packed Shamir k={k} t={t} n={n}, {P62.bit_length()}-bit prime {P62}; {rows} sealed boxes per tile",
    "box_bytes_per_tile": box_bytes, "box_bytes_per_secret": box_bytes / (P * dim),
    "ms_per_tile": ms.value / tiles, "elements_per_s": elements / (ms.value * 1e-3),
    "stage_ms_per_tile": {s: v / tiles for s, v in stage_ms.items()},
    "whole_config3_job_s": 100_000 * dim / (elements / (ms.value * 1e-3)),
    "verified_reveal_equals_sum_of_secrets": ok}, indent=1))
