"""Parity tests of the batched sealed-box open / seal (reference: encryption/sodium.rs:43, :78)
through the C ABI against oracle/sealedbox_oracle.py and the committed fixtures: byte-exact boxes with injected
ephemeral keys, oracle-sealed payloads opened on the device, tamper rejection ("Sodium decryption failure"), and the
clerk's whole job - open -> varint decode -> clerk sums - without leaving HBM."""
import random

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
h = bytes.fromhex
P62 = 4611686006577364993


def test_fixture_vectors_seal_and_open_host_forms(gpu):
    from sda_amd import capi, crypto
    box = crypto.SealedBox()
    for v in load_golden("sealedbox.json")["sealed"]["vectors"]:
        pk, sk, esk, m, c = (h(v[k]) for k in ("pk", "sk", "esk", "m", "c"))
        assert box.seal(m, pk, esk) == c                      # bit for bit with the injected ephemeral key
        assert box.open(c, pk, sk) == m
        for pos in {0, 31, 32, 47, len(c) - 1}:
            bad = bytearray(c); bad[pos] ^= 0x80
            with pytest.raises(capi.SdaError) as e:
                box.open(bytes(bad), pk, sk)
            assert e.value.code == capi.ERR_SODIUM_DECRYPTION and "Sodium decryption failure" in str(e.value)
    for short in (b"", bytes(47)):
        with pytest.raises(capi.SdaError) as e:
            box.open(short, pk, sk)
        assert e.value.code == capi.ERR_SODIUM_DECRYPTION
    # OS-entropy ephemeral keys: two seals of one message differ, both open
    a, b = box.seal(m, pk), box.seal(m, pk)
    assert a != b and box.open(a, pk, sk) == m and box.open(b, pk, sk) == m


def test_nacl_paper_box_through_the_kernels(gpu):
    """the worked example of "Cryptography in NaCl": Alice's secret key as the ephemeral key of a box to Bob gives the
    paper's ciphertext only under the paper's nonce - the sealed-box nonce differs - so this checks the pieces the
    fixture cannot: a seal to Bob with esk = Alice opens with Bob's key, and epk is Alice's public key"""
    from sda_amd import crypto
    from oracle import sealedbox_oracle as so
    k = load_golden("sealedbox.json")["kats"]
    alice_sk, alice_pk = h(k["x25519_base"][0]["scalar"]), h(k["x25519_base"][0]["out"])
    bob_sk, bob_pk = h(k["x25519_base"][1]["scalar"]), h(k["x25519_base"][1]["out"])
    m = h(k["secretbox"][0]["m"])
    c = crypto.SealedBox().seal(m, bob_pk, alice_sk)
    assert c[:32] == alice_pk and c == so.seal(m, bob_pk, alice_sk)
    assert crypto.SealedBox().open(c, bob_pk, bob_sk) == m


@pytest.mark.parametrize("lens", [[0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100],
                                  [16383, 16384, 16385, 16415, 16416, 16417, 32768, 40000],     # around the 16 KiB Poly1305 regions
                                  [1_000_003, 5, 262_144 + 32]])
def test_rows_open_vs_oracle_and_seal_vs_oracle(gpu, lens):
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import sealedbox_oracle as so
    rng = random.Random(sum(lens))
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n)) if n < 4096 else np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8).tobytes()
    rows = len(lens)
    sk = rb(32); pk = so.x25519_base(sk)
    msgs = [rb(n) for n in lens]
    esks = [rb(32) for _ in range(rows)]
    boxes = [so.seal(m, pk, e) for m, e in zip(msgs, esks)]
    job = crypto.JobContainer.build(0, boxes)
    L = job.layout
    d_job = DeviceBytes.from_bytes(bytes(job))
    out_slot = (max(lens) + 15) // 16 * 16 + 16
    d_out = DeviceBytes(rows * out_slot).zero()
    d_nb, d_ok, d_status = DeviceBytes(rows * 8).zero(), DeviceBytes(rows * 4).zero(), DeviceBytes(4).zero()
    box = crypto.SealedBox()
    box.open_rows_dev(pk, sk, d_job.ptr + L.payload_offset, L.slot_bytes, d_job.ptr + L.lengths_offset, rows, max(lens) + 48,
                      d_out.ptr, out_slot, d_nb.ptr, d_status.ptr, d_ok.ptr)
    ob = d_out.to_bytes()
    assert d_status.to_bytes() == bytes(4)
    assert list(np.frombuffer(d_ok.to_bytes(), dtype="<u4")) == [1] * rows
    assert list(np.frombuffer(d_nb.to_bytes(), dtype="<u8")) == lens
    for r in range(rows):
        assert ob[r * out_slot:r * out_slot + lens[r]] == msgs[r], f"row {r}"
    # seal on the device with the same ephemeral keys: the boxes must equal the oracle's, byte for byte
    mslot = out_slot
    blob = bytearray(rows * mslot)
    for r, m in enumerate(msgs):
        blob[r * mslot:r * mslot + len(m)] = m
    d_msgs = DeviceBytes.from_bytes(blob)
    d_mlen = DeviceBytes.from_bytes(np.array(lens, dtype="<u8").tobytes())
    bslot = (max(lens) + 48 + 15) // 16 * 16
    d_boxes, d_blen = DeviceBytes(rows * bslot).zero(), DeviceBytes(rows * 8).zero()
    box.seal_rows_dev([pk], 1, d_msgs.ptr, mslot, d_mlen.ptr, rows, max(lens), d_boxes.ptr, bslot, d_blen.ptr, esk=b"".join(esks))
    bb = d_boxes.to_bytes()
    assert list(np.frombuffer(d_blen.to_bytes(), dtype="<u8")) == [n + 48 for n in lens]
    for r in range(rows):
        assert bb[r * bslot:r * bslot + lens[r] + 48] == boxes[r], f"row {r}"


def test_rows_tamper_and_short_rows(gpu):
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import sealedbox_oracle as so
    rng = random.Random(3)
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    sk = rb(32); pk = so.x25519_base(sk)
    msgs = [rb(rng.randrange(0, 3000)) for _ in range(40)]
    boxes = [bytearray(so.seal(m, pk, rb(32))) for m in msgs]
    verdict = []
    for r, b in enumerate(boxes):
        kind = r % 5
        if kind == 1: b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)          # a flipped bit anywhere
        if kind == 2: del b[-1]                                                   # truncated
        if kind == 3: boxes[r] = bytearray(rb(rng.randrange(0, 48)))              # shorter than a box
        verdict.append(kind in (0, 4) or (kind == 2 and False))
    boxes = [bytes(b) for b in boxes]
    for r, b in enumerate(boxes):                                                 # the oracle agrees with the planned verdicts
        try:
            ok = so.seal_open(b, pk, sk) == msgs[r]
        except ValueError:
            ok = False
        assert ok == verdict[r], r
    job = crypto.JobContainer.build(0, boxes)
    L = job.layout
    d_job = DeviceBytes.from_bytes(bytes(job))
    rows, out_slot = len(boxes), 3072
    d_out, d_nb = DeviceBytes.from_bytes(b"\xC3" * (rows * out_slot)), DeviceBytes(rows * 8).zero()
    d_ok, d_status = DeviceBytes(rows * 4).zero(), DeviceBytes(4).zero()
    crypto.SealedBox().open_rows_dev(pk, sk, d_job.ptr + L.payload_offset, L.slot_bytes, d_job.ptr + L.lengths_offset, rows,
                                     max(len(b) for b in boxes), d_out.ptr, out_slot, d_nb.ptr, d_status.ptr, d_ok.ptr)
    ok = np.frombuffer(d_ok.to_bytes(), dtype="<u4")
    nb = np.frombuffer(d_nb.to_bytes(), dtype="<u8")
    assert [bool(x) for x in ok] == verdict
    assert np.frombuffer(d_status.to_bytes(), dtype="<u4")[0] == 16              # "Sodium decryption failure" for the job
    ob = d_out.to_bytes()
    for r in range(rows):
        assert nb[r] == (len(msgs[r]) if verdict[r] else 0)
        if verdict[r]:
            assert ob[r * out_slot:r * out_slot + len(msgs[r])] == msgs[r]
        else:
            # verify-then-decrypt: the slot of a box that fails is exactly what the caller passed - no unauthenticated
            # plaintext of a forged box ever lands in d_out
            assert ob[r * out_slot:(r + 1) * out_slot] == b"\xC3" * out_slot, r


def test_more_rows_than_one_launch_slice(gpu):
    """70,000 small boxes: the bulk kernels are issued in slices of 65,535 rows (grid.y); every row must still open, and
    rows on both sides of the slice boundary match the oracle"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import sealedbox_oracle as so
    rows, mlen = 70_000, 100
    rng = np.random.default_rng(70)
    sk = bytes(rng.integers(0, 256, 32, dtype=np.uint8)); pk = so.x25519_base(sk)
    msgs = rng.integers(0, 256, size=(rows, mlen), dtype=np.uint8)
    mslot, bslot = 112, 160
    blob = np.zeros((rows, mslot), dtype=np.uint8); blob[:, :mlen] = msgs
    d_msgs = DeviceBytes.from_bytes(blob.tobytes())
    d_mlen = DeviceBytes.from_bytes(np.full(rows, mlen, dtype="<u8").tobytes())
    d_boxes, d_blen = DeviceBytes(rows * bslot).zero(), DeviceBytes(rows * 8).zero()
    box = crypto.SealedBox()
    esk = rng.integers(0, 256, size=(rows, 32), dtype=np.uint8)
    box.seal_rows_dev([pk], rows, d_msgs.ptr, mslot, d_mlen.ptr, rows, mlen, d_boxes.ptr, bslot, d_blen.ptr, esk=esk.tobytes())
    bb = d_boxes.to_bytes()
    for r in (0, 1, 65534, 65535, 65536, rows - 1):
        assert bb[r * bslot:r * bslot + mlen + 48] == so.seal(msgs[r].tobytes(), pk, esk[r].tobytes()), r
    d_out, d_nb = DeviceBytes(rows * mslot).zero(), DeviceBytes(rows * 8).zero()
    d_ok, d_status = DeviceBytes(rows * 4).zero(), DeviceBytes(4).zero()
    box.open_rows_dev(pk, sk, d_boxes.ptr, bslot, d_blen.ptr, rows, mlen + 48, d_out.ptr, mslot, d_nb.ptr, d_status.ptr, d_ok.ptr)
    assert d_status.to_bytes() == bytes(4)
    assert np.frombuffer(d_ok.to_bytes(), dtype="<u4").all()
    out = np.frombuffer(d_out.to_bytes(), dtype=np.uint8).reshape(rows, mslot)
    assert np.array_equal(out[:, :mlen], msgs)


def test_small_order_ephemeral_keys_are_refused(gpu):
    """an ephemeral public key of small order makes the X25519 result all-zero, i.e. a key anybody can compute; libsodium's
    scalar multiplication reports that and the box does not open - here too, even though its tag verifies under that key"""
    from sda_amd import capi, crypto
    from oracle import sealedbox_oracle as so
    sk = bytes(range(7, 39)); pk = so.x25519_base(sk)
    m = b"forged without knowing sk"
    for epk in (bytes(32), (1).to_bytes(32, "little"),
                bytes.fromhex("e0eb7a7c3b41b8ae1656e3faf19fc46ada098deb9c32b1fd866205165f49b800")):   # orders 4, 1, 8
        assert so.x25519(sk, epk) == bytes(32)
        forged = epk + so.secretbox(m, so.seal_nonce(epk, pk), so.hsalsa20(bytes(32), bytes(16)))
        with pytest.raises(ValueError):
            so.seal_open(forged, pk, sk)
        with pytest.raises(capi.SdaError) as e:
            crypto.SealedBox().open(forged, pk, sk)
        assert e.value.code == capi.ERR_SODIUM_DECRYPTION


def test_sealing_to_a_small_order_key_is_refused_and_public_key(gpu):
    """crypto_box_seal returns -1 for a recipient key of small order (all-zero shared secret: anybody could open the box).
    Host form: an error and a wiped buffer; rows form: length 0 for that row, no tag, the payload NOT encrypted under the
    degenerate key, the other rows sealed as usual.  Also sda_sealedbox_public_key == X25519(sk, 9) of the oracle."""
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBytes
    from oracle import sealedbox_oracle as so
    box = crypto.SealedBox()
    for sk in (bytes(range(32)), bytes(range(7, 39)), b"\xff" * 32):
        assert box.public_key(sk) == so.x25519_base(sk)
    small = [bytes(32), (1).to_bytes(32, "little"),
             bytes.fromhex("e0eb7a7c3b41b8ae1656e3faf19fc46ada098deb9c32b1fd866205165f49b800")]
    for pk in small:
        with pytest.raises(capi.SdaError) as e:
            box.seal(b"secret shares", pk, bytes(range(1, 33)))
        assert e.value.code == capi.ERR_INVALID_ARGUMENT and "small-order" in str(e.value)
    good_sk = bytes(range(9, 41)); good_pk = so.x25519_base(good_sk)
    msg = bytes(range(200))
    rows, mslot, bslot = 4, 208, 256
    pks = [good_pk, small[0], good_pk, small[2]]
    esk = b"".join(bytes([i + 1]) * 32 for i in range(rows))
    d_msgs = DeviceBytes.from_bytes(b"".join(msg + bytes(mslot - len(msg)) for _ in range(rows)))
    d_mb = DeviceBytes.from_bytes(np.full(rows, len(msg), dtype="<u8").tobytes())
    d_boxes, d_rb = DeviceBytes.from_bytes(b"\x3C" * (rows * bslot)), DeviceBytes(rows * 8).zero()
    box.seal_rows_dev(pks, 1, d_msgs.ptr, mslot, d_mb.ptr, rows, len(msg), d_boxes.ptr, bslot, d_rb.ptr, esk=esk)
    rb = np.frombuffer(d_rb.to_bytes(), dtype="<u8")
    bb = d_boxes.to_bytes()
    assert list(rb) == [len(msg) + 48, 0, len(msg) + 48, 0]
    for r in range(rows):
        got = bb[r * bslot:(r + 1) * bslot]
        if r in (0, 2):
            assert got[:len(msg) + 48] == so.seal(msg, good_pk, esk[32 * r:32 * r + 32])
        else:
            assert got[32:] == b"\x3C" * (bslot - 32)              # no tag, no ciphertext: only the epk was written


def test_rows_longer_than_the_declared_bound_are_refused_not_read(gpu):
    """a row whose length field exceeds max_box_bytes (a lying job header) fails closed: flagged, length 0, and the
    kernels never touch bytes past the bound (the rows behind it stay intact and still open)"""
    from sda_amd import crypto
    from sda_amd.device import DeviceBytes
    from oracle import sealedbox_oracle as so
    rng = random.Random(12)
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    sk = rb(32); pk = so.x25519_base(sk)
    msgs = [rb(100), rb(200), rb(50)]
    boxes = [so.seal(m, pk, rb(32)) for m in msgs]
    slot = 256
    blob = bytearray(3 * slot)
    for r, b in enumerate(boxes):
        blob[r * slot:r * slot + len(b)] = b
    lens = np.array([len(boxes[0]), 1 << 40, len(boxes[2])], dtype="<u8")          # row 1 claims a terabyte
    d_boxes, d_lens = DeviceBytes.from_bytes(blob), DeviceBytes.from_bytes(lens.tobytes())
    d_out, d_nb = DeviceBytes(3 * 256).zero(), DeviceBytes(24).zero()
    d_ok, d_status = DeviceBytes(12).zero(), DeviceBytes(4).zero()
    crypto.SealedBox().open_rows_dev(pk, sk, d_boxes.ptr, slot, d_lens.ptr, 3, 248, d_out.ptr, 256, d_nb.ptr, d_status.ptr, d_ok.ptr)
    assert list(np.frombuffer(d_ok.to_bytes(), dtype="<u4")) == [1, 0, 1]
    assert list(np.frombuffer(d_nb.to_bytes(), dtype="<u8")) == [100, 0, 50]
    ob = d_out.to_bytes()
    assert ob[:100] == msgs[0] and ob[512:562] == msgs[2] and ob[256:512] == bytes(256)
    assert np.frombuffer(d_status.to_bytes(), dtype="<u4")[0] == 16


def test_clerking_job_open_decode_sum_on_the_device(gpu):
    """clerk.rs:78-86 with the job resident in HBM: P sealed share vectors (participate.rs:82-101: varint + seal per
    clerk) -> batch open -> streaming varint clerk sums; equals the oracle's combine of the plaintext shares.  The
    sealing side runs on the device too (seal_rows_dev over the slotted varint rows of sda_varint_encode_rows_dev)."""
    from sda_amd import capi, crypto
    from sda_amd.device import DeviceBuffer, DeviceBytes
    from oracle import coracle, sealedbox_oracle as so
    rng = np.random.default_rng(8)
    P, L = 64, 5000
    shares = rng.integers(0, P62, size=(P, L), dtype=np.int64)
    sk = bytes(rng.integers(0, 256, 32, dtype=np.uint8)); pk = so.x25519_base(sk)
    codec, box = crypto.VarintCodec(), crypto.SealedBox()
    d_sh = DeviceBuffer.from_numpy(shares)
    vslot = codec.slot_size(L)
    d_wire, d_wlen = DeviceBytes(P * vslot).zero(), DeviceBytes(P * 8).zero()
    codec.encode_rows_dev(d_sh.ptr, P, L, L, d_wire.ptr, vslot, d_wlen.ptr)
    bslot = vslot + 48
    d_boxes, d_blen = DeviceBytes(P * bslot).zero(), DeviceBytes(P * 8).zero()
    box.seal_rows_dev([pk], P, d_wire.ptr, vslot, d_wlen.ptr, P, vslot, d_boxes.ptr, bslot, d_blen.ptr)    # OS-entropy ephemeral keys
    # the boxes are what the reference would have produced for these payloads: the oracle opens them
    bb, bl = d_boxes.to_bytes(), np.frombuffer(d_blen.to_bytes(), dtype="<u8")
    for p in (0, P - 1):
        assert so.seal_open(bb[p * bslot:p * bslot + int(bl[p])], pk, sk) == coracle.varint_encode(shares[p])
    # clerk side
    d_plain, d_plen = DeviceBytes(P * vslot).zero(), DeviceBytes(P * 8).zero()
    d_status = DeviceBytes(4).zero()
    box.open_rows_dev(pk, sk, d_boxes.ptr, bslot, d_blen.ptr, P, bslot, d_plain.ptr, vslot, d_plen.ptr, d_status.ptr)
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    comb.begin_dev(1, L)
    comb.update_encoded_rows_dev(codec, d_plain.ptr, vslot, d_plen.ptr, P, d_status.ptr)
    sums = DeviceBuffer(L)
    comb.finish_dev(sums.ptr)
    assert d_status.to_bytes() == bytes(4)
    assert np.array_equal(sums.to_numpy(), coracle.combine(P62, shares))
    # trait-shaped host forms (ShareEncryptor::encrypt / ShareDecryptor::decrypt)
    enc, dec = crypto.ShareEncryptor(pk), crypto.ShareDecryptor(pk, sk)
    assert np.array_equal(dec.decrypt(enc.encrypt(shares[0][:100])), shares[0][:100])
