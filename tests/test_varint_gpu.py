"""Parity tests of the share-vector wire codec through the C ABI, against the oracle's
restatement of integer-encoding 1.0 `VarInt for i64` (sodium.rs:36-41, :83-89).  Byte-exact."""
import numpy as np
import pytest

from conftest import set_knob

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


@pytest.fixture(params=["scan", "stream"])
def decode_path(request, monkeypatch):
    """the two decode forms: three-pass block scan (few rows) and single-pass row streaming (many rows)"""
    set_knob("SDA_VARINT_PATH", request.param)
    return request.param


@pytest.mark.parametrize("n,kind", [(1, "edge"), (7, "small"), (2047, "shares"), (2048, "shares"), (2049, "mixed"),
                                    (100_003, "mixed"), (1_000_000, "shares"), (300_000, "signed")])
def test_encode_decode_vs_oracle(gpu, n, kind, decode_path):
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


def test_malformed_streams_are_refused(gpu, decode_path):
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


def test_device_rows_roundtrip_feeds_the_combiner(gpu, decode_path):
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


def _mixed(rng, n):
    bits = rng.integers(0, 64, size=n)
    return (rng.integers(-(2 ** 63), 2 ** 63 - 1, size=n, dtype=np.int64) >> (63 - bits)).astype(np.int64)


@pytest.mark.parametrize("rows,L,kind,shift", [(1, 1, "mixed", 0), (3, 7, "mixed", 1), (70, 113, "mixed", 5), (9, 5001, "shares", 0),
                                               (5, 3000, "tiny", 3), (4, 2500, "huge", 15), (2100, 257, "mixed", 0),
                                               (33, 1024, "tiny", 0)])
def test_row_streaming_decode(gpu, monkeypatch, rows, L, kind, shift):
    """single-pass row streaming == three-pass scan == oracle, for every value length, rows that start at any
    byte alignment (also a byte stream that itself starts off the 16-byte grid), 1-byte values (1024 per chunk)
    and 10-byte values; the status flags of damaged rows agree too."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(rows * 1000 + L)
    if kind == "mixed":
        v = _mixed(rng, rows * L).reshape(rows, L)
    elif kind == "shares":
        v = rng.integers(0, P62, size=(rows, L), dtype=np.int64)
    elif kind == "tiny":
        v = rng.integers(-64, 64, size=(rows, L), dtype=np.int64)               # one byte each
    else:
        v = rng.integers(2 ** 62, 2 ** 63 - 1, size=(rows, L), dtype=np.int64) * rng.choice([-1, 1], size=(rows, L))
    enc = [coracle.varint_encode(r) for r in v]
    raw = b"".join(enc)
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
    host = b"\xff" * shift + raw                                                # the stream starts `shift` bytes in
    d_bytes = DeviceBuffer.from_numpy(np.frombuffer(host + b"\0" * (-len(host) % 8), dtype=np.int64))
    d_off = DeviceBuffer.from_numpy(offs)
    codec = crypto.VarintCodec()
    stride = L + 3
    got = {}
    for path in ("scan", "stream"):
        set_knob("SDA_VARINT_PATH", path)
        d_dec = DeviceBuffer(rows * stride).zero()
        st = DeviceBuffer(1).zero()
        codec.decode_dev(d_bytes.ptr + shift, len(raw), d_off.ptr, rows, L, d_dec.ptr, stride, st.ptr)
        assert st.to_numpy()[0] == 0, path
        m = d_dec.to_numpy().reshape(rows, stride)
        assert np.array_equal(m[:, :L], v), path
        assert not m[:, L:].any(), path                                         # nothing written past a row
        got[path] = m
    # damaged rows: same verdicts from both forms
    def status(path, raw_b, offs_b, want_len):
        set_knob("SDA_VARINT_PATH", path)
        db = DeviceBuffer.from_numpy(np.frombuffer(raw_b + b"\0" * (-len(raw_b) % 8 or 8), dtype=np.int64))
        do = DeviceBuffer.from_numpy(np.asarray(offs_b, dtype=np.int64))
        st = DeviceBuffer(1).zero()
        dd = DeviceBuffer(rows * stride + 8).zero()
        codec.decode_dev(db.ptr, len(raw_b), do.ptr, rows, want_len, dd.ptr, max(stride, want_len), st.ptr)
        return int(st.to_numpy()[0])
    cases = [(raw, offs, L + 1), (raw, offs, max(L - 1, 0))]                     # wrong expected length
    if L > 1:
        bad = bytearray(raw); bad[int(offs[rows // 2 + 1]) - 1] |= 0x80          # a row that ends inside a value
        cases.append((bytes(bad), offs, L))
        moved = offs.copy(); moved[rows // 2 + 1 if rows > 1 else 0] += 0        # unchanged: control
        cases.append((raw, moved, L))
    if kind == "huge":
        over = bytearray(raw); over[3] |= 0x80; over[9] |= 0x80                  # >= 11 bytes without a terminator
        cases.append((bytes(over), offs, L))
    for raw_b, offs_b, want in cases:
        a, b = status("scan", raw_b, offs_b, want), status("stream", raw_b, offs_b, want)
        assert (a != 0) == (b != 0), (a, b, want)
        if a:
            assert a & b, (a, b, want)                                           # at least one common reason


@pytest.mark.parametrize("jobs,rpj,L,kind", [(1, 40, 5001, "shares"), (3, 17, 3000, "mixed"), (2, 16, 9000, "tiny"),
                                             (1, 33, 2500, "signed"), (8, 5, 1, "mixed"), (1, 1, 70_000, "shares"),
                                             (2, 35, 4097, "drift")])
def test_wire_format_clerk_sums(gpu, monkeypatch, jobs, rpj, L, kind):
    """clerk.rs:78-86 on wire-format rows, job-major: the single-pass form (rows streamed in lockstep, sums in an LDS
    column window, no decoded tile) == decode-then-combine == the oracle, bit-exact.  'tiny' overflows the window in
    one step (1024 one-byte values per chunk), 'drift' gives every row its own value size so the rows run apart, both
    exercising the direct-to-accumulator path; two updates check that the window state does not leak between calls."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(jobs * 100 + rpj)
    rows = jobs * rpj
    if kind == "shares":
        v = rng.integers(0, P62, size=(rows, L), dtype=np.int64)
    elif kind == "signed":
        v = rng.integers(-(P62 - 1), P62, size=(rows, L), dtype=np.int64)
    elif kind == "tiny":
        v = rng.integers(-64, 64, size=(rows, L), dtype=np.int64)
    elif kind == "drift":
        v = np.stack([rng.integers(0, 2 ** int(rng.integers(3, 62)), size=L, dtype=np.int64) for _ in range(rows)])
    else:
        v = _mixed(rng, rows * L).reshape(rows, L)
    enc = [coracle.varint_encode(r) for r in v]
    raw = b"".join(enc)
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
    d_bytes = DeviceBuffer.from_numpy(np.frombuffer(raw + b"\0" * (-len(raw) % 8 or 8), dtype=np.int64))
    d_off = DeviceBuffer.from_numpy(offs)
    codec = crypto.VarintCodec()
    q = P62
    want = np.stack([coracle.combine(q, np.concatenate([v[j * rpj:(j + 1) * rpj]] * 2)) for j in range(jobs)])
    for path in ("scan", "stream"):
        set_knob("SDA_VARINT_PATH", path)
        comb = crypto.ShareCombiner(crypto.Additive(3, q))
        st = DeviceBuffer(1).zero()
        out = DeviceBuffer(jobs * L)
        comb.begin_dev(jobs, L)
        comb.update_encoded_dev(codec, d_bytes.ptr, len(raw), d_off.ptr, rows, st.ptr)
        comb.update_encoded_dev(codec, d_bytes.ptr, len(raw), d_off.ptr, rows, st.ptr)
        comb.finish_dev(out.ptr)
        assert st.to_numpy()[0] == 0, path
        assert np.array_equal(out.to_numpy().reshape(jobs, L), want), path
    # a damaged row is reported by both forms
    if L > 1:
        bad = bytearray(raw); bad[int(offs[rows // 2 + 1]) - 1] |= 0x80
        db = DeviceBuffer.from_numpy(np.frombuffer(bytes(bad) + b"\0" * (-len(raw) % 8 or 8), dtype=np.int64))
        for path in ("scan", "stream"):
            set_knob("SDA_VARINT_PATH", path)
            comb = crypto.ShareCombiner(crypto.Additive(3, q))
            st = DeviceBuffer(1).zero()
            comb.begin_dev(jobs, L)
            comb.update_encoded_dev(codec, db.ptr, len(raw), d_off.ptr, rows, st.ptr)
            assert st.to_numpy()[0] != 0, path


@pytest.mark.parametrize("rows,L,kind,stride_pad", [(1, 1, "mixed", 0), (5, 127, "mixed", 1), (9, 128, "mixed", 0), (7, 129, "mixed", 3),
                                                    (33, 5001, "shares", 0), (6, 3000, "tiny", 2), (4, 2500, "huge", 0),
                                                    (40, 4096, "signed", 1), (3, 0, "mixed", 0), (2100, 61, "mixed", 0)])
def test_slotted_rows_encode_decode_combine(gpu, rows, L, kind, stride_pad):
    """single-pass slotted encode: the bytes of every row equal the oracle's encoding of that vector (all value
    lengths, odd strides -> unaligned rows take the scalar loads); decoded back bit-exactly; summed straight from
    the slots == the oracle's combine."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(rows * 7919 + L)
    if kind == "mixed":
        v = _mixed(rng, rows * L).reshape(rows, L)
    elif kind == "shares":
        v = rng.integers(0, P62, size=(rows, L), dtype=np.int64)
    elif kind == "signed":
        v = rng.integers(-(P62 - 1), P62, size=(rows, L), dtype=np.int64)
    elif kind == "tiny":
        v = rng.integers(-64, 64, size=(rows, L), dtype=np.int64)
    else:
        v = rng.integers(2 ** 62, 2 ** 63 - 1, size=(rows, L), dtype=np.int64) * rng.choice([-1, 1], size=(rows, L))
    if L:
        v[0, :min(L, 20)] = _edge_values()[:min(L, 20)]
    stride = L + stride_pad
    vals = np.zeros((rows, max(stride, 1)), dtype=np.int64)
    vals[:, :L] = v
    d_vals = DeviceBuffer.from_numpy(vals)
    codec = crypto.VarintCodec()
    slot = codec.slot_size(L) + 16
    assert slot % 16 == 0 and slot >= 10 * L
    d_bytes = DeviceBuffer(rows * slot // 8 + 2)
    d_len = DeviceBuffer(rows).zero()
    codec.encode_rows_dev(d_vals.ptr, rows, L, max(stride, 1) if L == 0 else stride, d_bytes.ptr, slot, d_len.ptr)
    lens = d_len.to_numpy().astype(np.uint64)
    raw = d_bytes.to_numpy().view(np.uint8)
    for r in range(rows):
        want = coracle.varint_encode(v[r]) if L else b""
        assert int(lens[r]) == len(want), (r, int(lens[r]), len(want))
        assert raw[r * slot:r * slot + len(want)].tobytes() == want, r
    if L == 0:
        return
    stride2 = L + 2
    d_dec = DeviceBuffer(rows * stride2).zero()
    st = DeviceBuffer(1).zero()
    codec.decode_rows_dev(d_bytes.ptr, slot, d_len.ptr, rows, L, d_dec.ptr, stride2, st.ptr)
    assert st.to_numpy()[0] == 0
    assert np.array_equal(d_dec.to_numpy().reshape(rows, stride2)[:, :L], v)
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    out = DeviceBuffer(L)
    comb.begin_dev(1, L)
    comb.update_encoded_rows_dev(codec, d_bytes.ptr, slot, d_len.ptr, rows, st.ptr)
    comb.finish_dev(out.ptr)
    assert st.to_numpy()[0] == 0
    assert np.array_equal(out.to_numpy(), coracle.combine(P62, v))
    # a length that cuts a row short is reported
    bad = lens.astype(np.int64).copy(); bad[rows // 2] -= 1
    d_bad = DeviceBuffer.from_numpy(bad)
    st.zero()
    codec.decode_rows_dev(d_bytes.ptr, slot, d_bad.ptr, rows, L, d_dec.ptr, stride2, st.ptr)
    assert st.to_numpy()[0] != 0


def test_whole_pipeline_over_the_wire_format(gpu):
    """participants -> shares -> (slotted) wire bytes -> clerks' streaming sums -> reconstruct: every stage on the
    device, nothing decoded to a tile; the result is the sum of the secrets (participate.rs:75-76 ->
    sodium.rs:36-41 | :83-89 -> clerk.rs:85-86 -> receive.rs:140-152), and the clerk sums equal the oracle's combine
    of the oracle's own generated shares."""
    from sda_amd import crypto
    from sda_amd.capi import check
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    W8, W9 = 631229665360524489, 3451275676410824977
    KEY = bytes(range(32))
    k, t, n, dim, P = 3, 1, 8, 5000, 96
    sch = crypto.PackedShamir(k, n, t, P62, W8, W9)
    B = (dim + k - 1) // k
    Bs = (B + 15) // 16 * 16
    secrets = DeviceBuffer(P * dim)
    check(gpu.sda_fill_synthetic_dev(secrets.ptr, P, dim, dim, 0, 5, P62, None))
    gen = crypto.ShareGenerator(sch)
    gen.set_drbg_key(KEY)
    shares = DeviceBuffer(n * P * Bs)                                   # job-major: [clerk][participant][Bs]
    gen.generate_batch_dev(secrets.ptr, P, dim, dim, shares.ptr, Bs, P * Bs, first_participant=0)
    codec = crypto.VarintCodec()
    slot = codec.slot_size(B)
    wire = DeviceBuffer(n * P * slot // 8 + 2)
    lens = DeviceBuffer(n * P)
    codec.encode_rows_dev(shares.ptr, n * P, B, Bs, wire.ptr, slot, lens.ptr)      # one message per (clerk, participant)
    comb = crypto.ShareCombiner(sch)
    st = DeviceBuffer(1).zero()
    comb.begin_dev(n, B)
    comb.update_encoded_rows_dev(codec, wire.ptr, slot, lens.ptr, n * P, st.ptr)   # all clerks' jobs in one call
    sums = DeviceBuffer(n * B)
    comb.finish_dev(sums.ptr)
    assert st.to_numpy()[0] == 0
    S = sums.to_numpy().reshape(n, B)
    host = shares.to_numpy().reshape(n, P, Bs)[:, :, :B]
    assert np.array_equal(S, np.stack([coracle.combine(P62, host[c]) for c in range(n)]))
    # a participant's wire bytes are the oracle's encoding of the oracle's shares for the same draws
    sec = secrets.to_numpy().reshape(P, dim)
    rnd = coracle.drbg_fill(KEY, 3, B, t, P62)
    want = coracle.packed_generate_csprng(P62, k, t, n, W8, W9, sec[3], rnd, gen.csprng_share_map())
    raw = wire.to_numpy().view(np.uint8)
    ln = lens.to_numpy().astype(np.int64).reshape(n, P)
    for c in (0, 5):
        off = (c * P + 3) * slot
        assert raw[off:off + ln[c, 3]].tobytes() == coracle.varint_encode(want[c])
    rec = crypto.SecretReconstructor(sch, dim)
    idx = [6, 0, 3, 7]                                                  # any t + k clerks
    rows = DeviceBuffer.from_numpy(np.ascontiguousarray(S[idx]))
    out = DeviceBuffer(dim)
    rec.reconstruct_dev(idx, rows.ptr, B, B, out.ptr, dim)
    assert np.array_equal(out.to_numpy(), coracle.combine(P62, sec))


def _row_verdict(row: bytes, L: int):
    """(well_formed, values) of one row by the wire rules: ends on a terminator, no value longer than 10 bytes,
    exactly L values."""
    from oracle import coracle
    if L == 0:
        return len(row) == 0, np.zeros(0, dtype=np.int64)
    if not row or row[-1] & 0x80:
        return False, None
    b = np.frombuffer(row, dtype=np.uint8)
    ends = np.flatnonzero((b & 0x80) == 0)
    sizes = np.diff(np.concatenate([[-1], ends]))
    if sizes.max() > 10 or ends.size != L:
        return False, None
    return True, coracle.varint_decode(row)


def test_decoders_on_damaged_wire_data(gpu, monkeypatch):
    """the decoders parse bytes that untrusted participants produced: for mutated streams (bit flips, truncation,
    extra bytes, wrong row boundaries) both forms must reach the oracle's verdict, decode the sound ones exactly,
    and never write outside the rows they were given."""
    from sda_amd import crypto
    from sda_amd.device import DeviceBuffer
    from oracle import coracle
    rng = np.random.default_rng(2024)
    codec = crypto.VarintCodec()
    for case in range(120):
        rows, L = int(rng.integers(1, 6)), int(rng.integers(1, 260))
        v = _mixed(rng, rows * L).reshape(rows, L) if case % 3 else rng.integers(0, P62, size=(rows, L), dtype=np.int64)
        enc = [bytearray(coracle.varint_encode(r)) for r in v]
        kind = case % 6
        for _ in range(int(rng.integers(0, 4)) if kind else 0):
            e = enc[int(rng.integers(0, rows))]
            if kind == 1 and len(e):
                e[int(rng.integers(0, len(e)))] ^= 1 << int(rng.integers(0, 8))          # bit flip
            elif kind == 2 and len(e) > 1:
                del e[int(rng.integers(0, len(e))):]                                      # truncation
            elif kind == 3:
                e += bytes(rng.integers(0, 256, size=int(rng.integers(1, 14)), dtype=np.uint8))   # trailing bytes
            elif kind == 4 and len(e):
                pos = int(rng.integers(0, len(e)))
                e[pos:pos] = b"\x80" * int(rng.integers(1, 13))                           # a run of continuation bytes
            elif kind == 5 and len(e) > 2:
                e[int(rng.integers(0, len(e)))] = 0x00                                    # a spurious terminator
        raw = b"".join(bytes(e) for e in enc)
        offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
        verdicts = [_row_verdict(bytes(e), L) for e in enc]
        sound = all(ok for ok, _ in verdicts)
        d_bytes = DeviceBuffer.from_numpy(np.frombuffer(raw + b"\0" * (-len(raw) % 8 or 8), dtype=np.int64))
        d_off = DeviceBuffer.from_numpy(offs)
        stride = L + 4
        for path in ("scan", "stream"):
            set_knob("SDA_VARINT_PATH", path)
            guard = np.full((rows + 2, stride), -7, dtype=np.int64)                      # a guard row above and below
            d_out = DeviceBuffer.from_numpy(guard)
            st = DeviceBuffer(1).zero()
            codec.decode_dev(d_bytes.ptr, len(raw), d_off.ptr, rows, L, d_out.ptr + 8 * stride, stride, st.ptr)
            got = d_out.to_numpy().reshape(rows + 2, stride)
            assert (int(st.to_numpy()[0]) == 0) == sound, (case, path, int(st.to_numpy()[0]), sound)
            assert (got[0] == -7).all() and (got[-1] == -7).all() and (got[1:-1, L:] == -7).all(), (case, path)
            if sound:
                assert np.array_equal(got[1:-1, :L], np.stack([vals for _, vals in verdicts])), (case, path)
            # the clerk form on the same bytes: same verdict, and the exact sums when sound
            comb = crypto.ShareCombiner(crypto.Additive(3, P62))
            st.zero()
            out = DeviceBuffer(L)
            comb.begin_dev(1, L)
            comb.update_encoded_dev(codec, d_bytes.ptr, len(raw), d_off.ptr, rows, st.ptr)
            comb.finish_dev(out.ptr)
            assert (int(st.to_numpy()[0]) == 0) == sound, (case, path, "clerk")
            if sound:
                assert np.array_equal(out.to_numpy(), coracle.combine(P62, np.stack([vals for _, vals in verdicts])))
