"""CPU tests of the wire-level rows: the SDAJOBv1 container functions of the C ABI are host-only
byte layout code, so they run here without a GPU, against the independent restatement in oracle/wire_oracle.py; the
oracle's base64 is pinned by the RFC 4648 section 10 vectors and the strictness rules of data_encoding::base64::decode."""
import ctypes as C
import os
import random

import pytest

from oracle import wire_oracle as wo


def test_base64_oracle_rfc4648_vectors_and_strictness():
    for raw, text in wo.RFC4648_VECTORS:
        assert wo.binary_to_base64(raw) == text
        assert wo.binary_from_base64(text) == raw
    for bad in (b"Zg=", b"Zg", b"Z===", b"Zm9v\n", b"Zm9*", b"=Zm9", b"Zh==", b"Zm9=", b"Zg==Zg==", b"Zm9vY"):
        with pytest.raises(ValueError, match="Base64 decoding error"):
            wo.binary_from_base64(bad)


def test_job_container_matches_the_oracle_layout(built):
    from sda_amd import capi, crypto
    rng = random.Random(5)
    for rows, maxlen in [(0, 0), (1, 0), (1, 1), (3, 47), (7, 16), (100, 1000)]:
        payloads = [bytes(rng.randrange(256) for _ in range(rng.randrange(maxlen + 1))) for _ in range(rows)]
        for kind in (wo.SEALED, wo.VARINT, wo.BASE64_TEXT):
            jc = crypto.JobContainer.build(kind, payloads)
            want = wo.build_job(kind, payloads)
            assert bytes(jc)[:len(want)] == want                      # byte-identical blob
            k2, slot, back = wo.parse_job(bytes(jc))
            assert (k2, back) == (kind, payloads) and slot == jc.layout.slot_bytes
            assert crypto.JobContainer.parse(want).rows() == payloads
            L = jc.layout
            assert L.payload_offset % 16 == 0 and L.slot_bytes % 16 == 0 and L.lengths_offset == 64
            assert L.total_bytes == len(want) and L.rows == rows and L.payload_kind == kind
    # an explicit, larger slot
    jc = crypto.JobContainer.build(wo.VARINT, [b"abc", b""], slot_bytes=64)
    assert bytes(jc) == wo.build_job(wo.VARINT, [b"abc", b""], slot=64)


def test_job_container_refuses_malformed_headers(built):
    """a job is network input: every header field is checked, nothing is trusted"""
    from sda_amd import capi
    lib = capi.load()
    good = bytearray(wo.build_job(wo.SEALED, [b"x" * 40, b"y" * 7]))
    lay = capi.JobLayout()

    def parse(b):
        buf = (C.c_uint8 * len(b)).from_buffer(bytearray(b))
        return lib.sda_job_container_parse(buf, len(b), C.byref(lay))

    assert parse(good) == capi.OK
    cases = []
    for off, val in [(0, b"X"), (8, (63).to_bytes(4, "little")), (12, (3).to_bytes(4, "little")),
                     (16, (1 << 50).to_bytes(8, "little")), (24, (40).to_bytes(8, "little")),
                     (32, (72).to_bytes(8, "little")), (40, (96).to_bytes(8, "little")),
                     (48, (10**6).to_bytes(8, "little")), (56, (1).to_bytes(8, "little")),
                     (64, (49).to_bytes(8, "little"))]:                  # a row longer than its slot
        b = bytearray(good)
        b[off:off + len(val)] = val
        cases.append(bytes(b))
    cases.append(bytes(good[:-1]))                                        # truncated
    cases.append(bytes(good[:10]))
    for b in cases:
        assert parse(b) == capi.ERR_INVALID_ARGUMENT, b[:64]
        with pytest.raises(ValueError):
            wo.parse_job(b)
    # builders: slot not a multiple of 16, payload too long, row out of range
    buf = (C.c_uint8 * 4096)()
    assert lib.sda_job_container_size(3, 40) == 0
    assert lib.sda_job_container_init(buf, 4096, 0, 3, 40, None) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_job_container_init(buf, 100, 0, 3, 48, None) == capi.ERR_INVALID_ARGUMENT      # buffer too small
    assert lib.sda_job_container_init(buf, 4096, 0, 3, 48, None) == capi.OK
    assert lib.sda_job_container_set_row(buf, 4096, 3, b"a", 1) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_job_container_set_row(buf, 4096, 0, b"a" * 49, 49) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_job_container_set_row(buf, 4096, 2, b"a" * 48, 48) == capi.OK
    # set_row re-validates the header against the caller's capacity: a blob whose header claims more rows / a larger slot
    # than the buffer holds (a parsed, untrusted job) is refused before a byte is written
    total = lib.sda_job_container_size(3, 48)
    assert lib.sda_job_container_set_row(buf, total - 1, 2, b"a", 1) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_job_container_set_row(buf, total, 2, b"a", 1) == capi.OK
    lying = (C.c_uint8 * 4096).from_buffer_copy(bytes(buf))
    lying[16:24] = (1 << 30).to_bytes(8, "little")                        # rows: header no longer consistent with itself
    assert lib.sda_job_container_set_row(lying, 4096, 2, b"a", 1) == capi.ERR_INVALID_ARGUMENT
    # get_row is O(1): it checks the header and ITS row's length only - a corrupt length elsewhere does not stop it, the
    # corrupt row itself is refused, and the full parse still refuses the blob
    lo = 64
    bad = (C.c_uint8 * 4096).from_buffer_copy(bytes(buf))
    bad[lo + 8:lo + 16] = (49).to_bytes(8, "little")                      # row 1 claims 49 bytes in a 48-byte slot
    pay, ln = C.c_void_p(), C.c_size_t()
    assert lib.sda_job_container_get_row(bad, 4096, 2, C.byref(pay), C.byref(ln)) == capi.OK and ln.value == 1
    assert lib.sda_job_container_get_row(bad, 4096, 1, C.byref(pay), C.byref(ln)) == capi.ERR_INVALID_ARGUMENT
    lay = capi.JobLayout()
    assert lib.sda_job_container_parse(bad, 4096, C.byref(lay)) == capi.ERR_INVALID_ARGUMENT
    assert lib.sda_base64_encoded_size(0) == 0 and lib.sda_base64_encoded_size(1) == 4 and lib.sda_base64_encoded_size(3) == 4
    assert lib.sda_base64_decoded_max(8) == 6 and lib.sda_job_slot_size(17) == 32
