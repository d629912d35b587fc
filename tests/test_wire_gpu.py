"""Parity tests of the wire-level device code through the C ABI: base64 of `Binary` payloads
(protocol/src/helpers.rs:174-216) byte-exact against the oracle / Python's base64, the SDAJOBv1 container consumed in
place by the device calls, and the JSON-form job -> base64 decode -> streaming clerk sums pipeline."""
import ctypes as C
import json
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P62 = 4611686006577364993


def _u64(dev_bytes, count, offset=0):
    return np.frombuffer(dev_bytes.to_bytes(count * 8, offset), dtype="<u8")


@pytest.mark.parametrize("lens", [[0], [1], [2], [3], [4], [11, 12, 13], [47, 48, 49, 4095, 4096, 4097, 12289],
                                  [3_145_734, 0, 5, 3_000_001]])
def test_base64_rows_roundtrip_vs_oracle(gpu, lens):
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import wire_oracle as wo
    rng = np.random.default_rng(len(lens) * 1000 + lens[0])
    raws = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in lens]
    texts = [wo.binary_to_base64(r) for r in raws]
    rows, max_bytes = len(raws), max(lens)
    in_slot = (max_bytes + 15) // 16 * 16 + 16
    text_slot = (gpu.sda_base64_encoded_size(max_bytes) + 15) // 16 * 16 + 16
    blob = bytearray(rows * in_slot)
    for r, b in enumerate(raws):
        blob[r * in_slot:r * in_slot + len(b)] = b
    d_in = DeviceBytes.from_bytes(blob)
    d_in_bytes = DeviceBytes.from_bytes(np.array(lens, dtype="<u8").tobytes())
    d_text = DeviceBytes(rows * text_slot).zero()
    d_text_bytes = DeviceBytes(rows * 8).zero()
    crypto.base64_encode_rows_dev(d_in.ptr, in_slot, d_in_bytes.ptr, rows, max_bytes, d_text.ptr, text_slot, d_text_bytes.ptr)
    got_len = _u64(d_text_bytes, rows)
    tb = d_text.to_bytes()
    for r in range(rows):
        assert got_len[r] == len(texts[r])
        assert tb[r * text_slot:r * text_slot + len(texts[r])] == texts[r], f"row {r}"
    # and back
    d_out = DeviceBytes(rows * in_slot).zero()
    d_out_bytes = DeviceBytes(rows * 8).zero()
    d_status = DeviceBytes(4).zero()
    d_rs = DeviceBytes(rows * 4).zero()
    crypto.base64_decode_rows_dev(d_text.ptr, text_slot, d_text_bytes.ptr, rows, max(len(t) for t in texts), d_out.ptr, in_slot,
                                  d_out_bytes.ptr, d_status.ptr, d_rs.ptr)
    assert d_status.to_bytes() == bytes(4) and d_rs.to_bytes() == bytes(4 * rows)
    ob = d_out.to_bytes()
    assert list(_u64(d_out_bytes, rows)) == lens
    for r in range(rows):
        assert ob[r * in_slot:r * in_slot + lens[r]] == raws[r], f"row {r}"
        assert ob[r * in_slot + lens[r]:(r + 1) * in_slot] == bytes(in_slot - lens[r])        # nothing written past a row


def test_base64_more_rows_than_one_launch_slice(gpu):
    """70,000 short rows: the kernels are issued in slices of 65,535 rows; encode and decode round-trip for all of them"""
    import base64
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    rows, n = 70_000, 50
    rng = np.random.default_rng(1)
    raw = rng.integers(0, 256, size=(rows, n), dtype=np.uint8)
    in_slot, text_slot = 64, 80
    blob = np.zeros((rows, in_slot), dtype=np.uint8); blob[:, :n] = raw
    d_in = DeviceBytes.from_bytes(blob.tobytes())
    d_len = DeviceBytes.from_bytes(np.full(rows, n, dtype="<u8").tobytes())
    d_text, d_tlen = DeviceBytes(rows * text_slot).zero(), DeviceBytes(rows * 8).zero()
    crypto.base64_encode_rows_dev(d_in.ptr, in_slot, d_len.ptr, rows, n, d_text.ptr, text_slot, d_tlen.ptr)
    text = np.frombuffer(d_text.to_bytes(), dtype=np.uint8).reshape(rows, text_slot)
    tl = len(base64.b64encode(bytes(n)))
    assert (_u64(d_tlen, rows) == tl).all()
    for r in (0, 65534, 65535, 65536, rows - 1):
        assert text[r, :tl].tobytes() == base64.b64encode(raw[r].tobytes())
    d_out, d_olen = DeviceBytes(rows * in_slot).zero(), DeviceBytes(rows * 8).zero()
    d_status = DeviceBytes(4).zero()
    crypto.base64_decode_rows_dev(d_text.ptr, text_slot, d_tlen.ptr, rows, tl, d_out.ptr, in_slot, d_olen.ptr, d_status.ptr)
    assert d_status.to_bytes() == bytes(4) and (_u64(d_olen, rows) == n).all()
    out = np.frombuffer(d_out.to_bytes(), dtype=np.uint8).reshape(rows, in_slot)
    assert np.array_equal(out[:, :n], raw)


def test_base64_decode_inside_a_json_document_and_malformed_rows(gpu):
    """rows addressed by (offset, length) straight into the JSON text of a ClerkingJob - any alignment - and the strict
    decoder's verdicts (helpers.rs:183 "Base64 decoding error") row by row, equal to the oracle's"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import wire_oracle as wo
    rng = random.Random(9)
    good = [bytes(rng.randrange(256) for _ in range(n)) for n in (1, 2, 3, 31, 32, 33, 1000, 0)]
    strings = [wo.binary_to_base64(g).decode() for g in good]
    bad = ["Zg=", "Zg", "Z===", "Zm9v\n", "Zm9*", "=Zm9", "Zh==", "Zm9=", "Zg==Zg==", "Zm9vY", "Zm9vZm9vZm9vZm9vZm9vZm9v*m9v", "Zm=v"]
    strings += bad
    doc = json.dumps({"id": "x", "encryptions": [{"Sodium": s} for s in strings]})
    offs, lens, pos = [], [], 0
    for s in strings:
        pos = doc.index(json.dumps(s), pos) + 1
        offs.append(pos); lens.append(len(s)); pos += len(s)
    assert len({o & 3 for o in offs}) > 1                                    # rows start at every alignment
    rows = len(strings)
    d_doc = DeviceBytes.from_bytes(doc.encode())
    d_offs = DeviceBytes.from_bytes(np.array(offs, dtype="<u8").tobytes())
    d_lens = DeviceBytes.from_bytes(np.array(lens, dtype="<u8").tobytes())
    out_slot = 1024
    d_out = DeviceBytes(rows * out_slot).zero()
    d_nb = DeviceBytes(rows * 8).zero()
    d_status, d_rs = DeviceBytes(4).zero(), DeviceBytes(rows * 4).zero()
    crypto.base64_decode_rows_dev(d_doc.ptr, 0, d_lens.ptr, rows, max(lens), d_out.ptr, out_slot, d_nb.ptr, d_status.ptr, d_rs.ptr,
                                  d_text_offsets=d_offs.ptr)
    rs = np.frombuffer(d_rs.to_bytes(), dtype="<u4")
    nb = _u64(d_nb, rows)
    ob = d_out.to_bytes()
    for r, s in enumerate(strings):
        try:
            want = wo.binary_from_base64(s.encode())
        except ValueError:
            want = None
        assert (rs[r] != 0) == (want is None), (r, s)
        if want is not None:
            assert nb[r] == len(want) and ob[r * out_slot:r * out_slot + len(want)] == want
    assert np.frombuffer(d_status.to_bytes(), dtype="<u4")[0] == 8


def test_rows_longer_than_the_declared_bound_are_refused_not_read(gpu):
    """Row lengths are network input (a job blob's length table, a JSON scan).  A row whose length exceeds max_chars /
    max_bytes - including an absurd 2^40 - must be refused without one byte of it read or written: byte count 0, status
    bit 8 and the row's flag (decode); text length 0 (encode); every neighbour untouched and still decoded."""
    import base64
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    rng = np.random.default_rng(77)
    rows, n = 6, 48
    raw = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for _ in range(rows)]
    texts = [base64.b64encode(b) for b in raw]
    tl = len(texts[0])
    text_slot, out_slot = 64, 48
    blob = bytearray(rows * text_slot)
    for r, t in enumerate(texts):
        blob[r * text_slot:r * text_slot + tl] = t
    lens = [tl, tl + 4, tl, 1 << 40, tl, (1 << 64) - 4]                     # rows 1, 3 and 5 lie about their length
    d_text = DeviceBytes.from_bytes(blob)
    d_lens = DeviceBytes.from_bytes(np.array(lens, dtype="<u8").tobytes())
    d_out = DeviceBytes.from_bytes(b"\xA5" * (rows * out_slot))
    d_nb = DeviceBytes.from_bytes(np.full(rows, 7, dtype="<u8").tobytes())
    d_status, d_rs = DeviceBytes(4).zero(), DeviceBytes(rows * 4).zero()
    crypto.base64_decode_rows_dev(d_text.ptr, text_slot, d_lens.ptr, rows, tl, d_out.ptr, out_slot, d_nb.ptr, d_status.ptr, d_rs.ptr)
    rs = np.frombuffer(d_rs.to_bytes(), dtype="<u4")
    nb = _u64(d_nb, rows)
    ob = d_out.to_bytes()
    assert list(rs != 0) == [False, True, False, True, False, True]
    assert np.frombuffer(d_status.to_bytes(), dtype="<u4")[0] == 8
    for r in range(rows):
        if r in (1, 3, 5):
            assert nb[r] == 0 and ob[r * out_slot:(r + 1) * out_slot] == b"\xA5" * out_slot      # not a byte written
        else:
            assert nb[r] == n and ob[r * out_slot:(r + 1) * out_slot] == raw[r]
    # encode: a raw row longer than max_bytes
    in_slot, tslot = 48, 64
    d_in = DeviceBytes.from_bytes(b"".join(raw))
    blens = [n, n + 1, 1 << 40, n]
    d_blens = DeviceBytes.from_bytes(np.array(blens, dtype="<u8").tobytes())
    d_t = DeviceBytes.from_bytes(b"\x5A" * (4 * tslot))
    d_tl = DeviceBytes.from_bytes(np.full(4, 9, dtype="<u8").tobytes())
    crypto.base64_encode_rows_dev(d_in.ptr, in_slot, d_blens.ptr, 4, n, d_t.ptr, tslot, d_tl.ptr)
    tlen = _u64(d_tl, 4)
    tb = d_t.to_bytes()
    assert list(tlen) == [tl, 0, 0, tl]
    for r in range(4):
        if r in (1, 2):
            assert tb[r * tslot:(r + 1) * tslot] == b"\x5A" * tslot
        else:
            assert tb[r * tslot:r * tslot + tl] == texts[r]


def test_json_job_to_clerk_sums_on_the_device(gpu):
    """The clerk's side of the reference with the job kept in HBM: the JSON form of a ClerkingJob (base64 strings,
    resources.rs:128-139) -> SDAJOBv1 text container -> base64 decode into a VARINT container's slots -> streaming
    clerk sums (clerk.rs:78-86), equal to the oracle's combine of the decoded vectors."""
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBuffer, DeviceBytes
    from oracle import coracle, wire_oracle as wo
    rng = np.random.default_rng(4)
    P, L = 48, 777
    shares = rng.integers(0, P62, size=(P, L), dtype=np.int64)
    payloads = [coracle.varint_encode(shares[p]) for p in range(P)]       # what an opened sealed box holds (sodium.rs:36-41)
    texts = [wo.binary_to_base64(b) for b in payloads]
    job = crypto.JobContainer.build(capi.JOB_BASE64_TEXT, texts)
    assert bytes(job) == wo.build_job(wo.BASE64_TEXT, texts)
    d_job = DeviceBytes.from_bytes(bytes(job))
    Lt = job.layout
    codec = crypto.VarintCodec()
    slot = codec.slot_size(L)
    out = crypto.JobContainer.build(capi.JOB_VARINT, [b""] * P, slot_bytes=slot)
    d_out = DeviceBytes.from_bytes(bytes(out))
    Lo = out.layout
    d_status = DeviceBytes(4).zero()
    crypto.base64_decode_rows_dev(d_job.ptr + Lt.payload_offset, Lt.slot_bytes, d_job.ptr + Lt.lengths_offset, P,
                                  max(len(t) for t in texts), d_out.ptr + Lo.payload_offset, Lo.slot_bytes,
                                  d_out.ptr + Lo.lengths_offset, d_status.ptr)
    back = crypto.JobContainer.parse(d_out.to_bytes())
    assert back.rows() == payloads
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    comb.begin_dev(1, L)
    comb.update_encoded_rows_dev(codec, d_out.ptr + Lo.payload_offset, Lo.slot_bytes, d_out.ptr + Lo.lengths_offset, P, d_status.ptr)
    sums = DeviceBuffer(L)
    comb.finish_dev(sums.ptr)
    assert d_status.to_bytes() == bytes(4)
    assert np.array_equal(sums.to_numpy(), coracle.combine(P62, shares))
