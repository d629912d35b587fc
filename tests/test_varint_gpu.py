"""Parity tests of the share-vector wire codec through the C ABI, against the oracle's
restatement of integer-encoding 1.0 `VarInt for i64` (sodium.rs:36-41, :83-89).  Byte-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P62 = 4611686006577364993


def _edge_values():
    e = [0, 1, -1, 2, -2, 63, 64, -64, -65, 8191, 8192, -8192, -8193, 2 ** 31 - 1, -2 ** 31, 2 ** 62, P62 - 1,
         2 ** 63 - 1, -2 ** 63, -2 ** 63 + 1]
    for b in range(0, 64, 7):
        e += [2 ** b, 2 ** b - 1, -(2 ** b), -(2 ** b) - 1] if b < 63 else []
    return np.array(e, dtype=np.int64)


def test_published_vectors(gpu):
    """protobuf sint64 / LEB128 textbook vectors (zig-zag: 0->0, -1->1, 1->2, -2->3; 150 -> 96 01)."""
    from sda_amd import crypto
    c = crypto.VarintCodec()
    assert c.encode([0]) == b"\x00" and c.encode([-1]) == b"\x01" and c.encode([1]) == b"\x02" and c.encode([-2]) == b"\x03"
    assert c.encode([75]) == bytes([0x96, 0x01])
    assert c.encode([2147483647]) == bytes([0xfe, 0xff, 0xff, 0xff, 0x0f])
    assert c.encode([-2147483648]) == bytes([0xff, 0xff, 0xff, 0xff, 0x0f])
    assert c.encode([2 ** 63 - 1]) == bytes([0xfe] + [0xff] * 8 + [0x01])
    assert c.encode([-2 ** 63]) == bytes([0xff] * 9 + [0x01])
    assert c.encode([]) == b"" and c.decode(b"").size == 0


@pytest.mark.parametrize("n,kind", [(1, "edge"), (7, "small"), (2047, "shares"), (2048, "shares"), (2049, "mixed"),
                                    (100_003, "mixed"), (1_000_000, "shares"), (300_000, "signed")])
def test_encode_decode_vs_oracle(gpu, n, kind):
    from sda_amd import crypto
    from oracle import coracle
    rng = np.random.default_rng(n)
    if kind == "edge":
        v = _edge_values()
    elif kind == "small":
        v = rng.integers(-100, 100, size=n, dtype=np.int64)
    elif kind == "shares":                                  # canonical 62-bit residues: 9 bytes each, mostly
        v = rng.integers(0, P62, size=n, dtype=np.int64)
    elif kind == "signed":                                  # the reference's (-q, q) shares
        v = rng.integers(-(P62 - 1), P62, size=n, dtype=np.int64)
    else:                                                   # every length 1..10 mixed
        bits = rng.integers(0, 64, size=n)
        v = (rng.integers(-(2 ** 63), 2 ** 63 - 1, size=n, dtype=np.int64) >> (63 - bits)).astype(np.int64)
        v[::97] = _edge_values()[rng.integers(0, _edge_values().size, size=v[::97].size)]
    c = crypto.VarintCodec()
    enc = c.encode(v)
    assert enc == coracle.varint_encode(v)
    dec = c.decode(enc)
    assert np.array_equal(dec, v)
    assert np.array_equal(coracle.varint_decode(enc), v)


def test_pyoracle_agrees_on_small_case(gpu):
    from sda_amd import crypto
    from oracle import pyoracle as po
    v = _edge_values()
    c = crypto.VarintCodec()
    assert c.encode(v) == po.varint_encode(v.tolist())
    assert c.decode(po.varint_encode(v.tolist())).tolist() == po.varint_decode(po.varint_encode(v.tolist()))


def test_malformed_streams_are_refused(gpu):
    from sda_amd import capi, crypto
    c = crypto.VarintCodec()
    with pytest.raises(capi.SdaError):                      # ends inside a value
        c.decode(bytes([0x02, 0x80]))
    with pytest.raises(capi.SdaError):                      # 11 continuation bytes
        c.decode(bytes([0x80] * 11 + [0x01]))
    from oracle import coracle, pyoracle as po
    ten = bytes([0x80] * 9 + [0x01])                        # 10 bytes is the legal maximum
    assert c.decode(ten).tolist() == [2 ** 62] == po.varint_decode(ten) == coracle.varint_decode(ten).tolist()
    ten_hi = bytes([0xff] * 9 + [0x7f])                     # bits above the 64th are dropped, as in the reference
    assert c.decode(ten_hi).tolist() == po.varint_decode(ten_hi) == coracle.varint_decode(ten_hi).tolist()


def test_device_rows_roundtrip_feeds_the_combiner(gpu):
    """participants' share vectors for one clerk: encode on device -> per-row byte ranges (what would be
    sealed) -> decode on device -> clerk combine; equals combining the originals."""
    import ctypes as C
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(9)
    rows, L, stride = 37, 5001, 5008
    vals = np.zeros((rows, stride), dtype=np.int64)
    vals[:, :L] = rng.integers(0, P62, size=(rows, L), dtype=np.int64)
    vals[3, :L] = rng.integers(-50, 50, size=L)             # a short-encoding row
    d_vals = DeviceBuffer.from_numpy(vals)
    codec = crypto.VarintCodec()
    cap = rows * L * 10
    d_bytes = DeviceBuffer((cap + 7) // 8)
    d_off = DeviceBuffer(rows + 1)
    total = codec.encode_dev(d_vals.ptr, rows, L, stride, d_bytes.ptr, cap, d_off.ptr)
    off = d_off.to_numpy().astype(np.uint64)
    raw = d_bytes.to_numpy().view(np.uint8)[:total].tobytes()
    assert off[0] == 0 and off[-1] == total
    for r in (0, 3, rows - 1):
        assert raw[int(off[r]):int(off[r + 1])] == coracle.varint_encode(vals[r, :L])
    assert raw == b"".join(coracle.varint_encode(vals[r, :L]) for r in range(rows))
    # decode into a fresh matrix with a different stride, then combine on the device
    stride2 = 5002
    d_dec = DeviceBuffer(rows * stride2).zero()
    d_status = DeviceBuffer(1).zero()
    codec.decode_dev(d_bytes.ptr, total, d_off.ptr, rows, L, d_dec.ptr, stride2, d_status.ptr)
    assert d_status.to_numpy()[0] == 0
    dec = d_dec.to_numpy().reshape(rows, stride2)[:, :L]
    assert np.array_equal(dec, vals[:, :L])
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    out = DeviceBuffer(L)
    comb.begin_dev(1, L)
    comb.update_dev(d_dec.ptr, 0, rows, stride2)
    comb.finish_dev(out.ptr)
    assert np.array_equal(out.to_numpy(), coracle.combine(P62, vals[:, :L]))
    # a wrong expected length is reported, not silently accepted
    d_status.zero()
    codec.decode_dev(d_bytes.ptr, total, d_off.ptr, rows, L - 1, d_dec.ptr, stride2, d_status.ptr)
    assert d_status.to_numpy()[0] & 2


def test_streaming_clerk_over_wire_format(gpu):
    """clerk.rs:78-86 without materialising the decoded vectors: every participation's opened payload goes
    decode -> clerk-sum update -> discard; equals the reference flow decode-all-then-combine."""
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle, pyoracle as po
    rng = np.random.default_rng(21)
    P, L = 19, 1237
    q = 433
    shares = rng.integers(-(q - 1), q, size=(P, L), dtype=np.int64)           # the reference's (-q, q) shares
    payloads = [po.varint_encode(row.tolist()) for row in shares]             # what ShareEncryptor seals
    codec = crypto.VarintCodec()
    comb = crypto.ShareCombiner(crypto.Additive(3, q))
    comb.begin(L)
    for raw in payloads:
        comb.update_encoded(codec, raw)
    got = comb.finish(L)
    want = po.combine([po.varint_decode(raw) for raw in payloads], q, "rust_signed")
    assert got.tolist() == [v % q for v in want]
    assert np.array_equal(got, coracle.combine(q, shares))
    # a payload of the wrong length is refused like the reference's combiner ("Wrong dimension") and leaves
    # the running sums untouched
    comb.begin(L)
    comb.update_encoded(codec, payloads[0])
    with pytest.raises(capi.SdaError) as e:
        comb.update_encoded(codec, po.varint_encode(shares[1][:-1].tolist()))
    assert e.value.code == capi.ERR_WRONG_DIMENSION and "Wrong dimension" in e.value.message
    with pytest.raises(capi.SdaError):
        comb.update_encoded(codec, payloads[2][:-1] + b"\x80")
    assert np.array_equal(comb.finish(L), np.mod(shares[0], q))
    # device-resident batch form
    big = rng.integers(0, P62, size=(64, 4001), dtype=np.int64)
    raw = b"".join(coracle.varint_encode(r) for r in big)
    offs = np.cumsum([0] + [len(coracle.varint_encode(r)) for r in big]).astype(np.int64)
    d_bytes = DeviceBuffer.from_numpy(np.frombuffer(raw + b"\0" * (-len(raw) % 8), dtype=np.int64))
    d_off = DeviceBuffer.from_numpy(offs)
    st = DeviceBuffer(1).zero()
    c2 = crypto.ShareCombiner(crypto.Additive(3, P62))
    c2.begin_dev(1, 4001)
    c2.update_encoded_dev(codec, d_bytes.ptr, len(raw), d_off.ptr, 64, st.ptr)
    out = DeviceBuffer(4001)
    c2.finish_dev(out.ptr)
    assert st.to_numpy()[0] == 0 and np.array_equal(out.to_numpy(), coracle.combine(P62, big))
