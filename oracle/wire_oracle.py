"""CPU ORACLE for the wire-level rows of SURVEY.md 8f: the SDAJOBv1 clerking-job container and the base64 form of
`Binary` payloads.  TEST INFRASTRUCTURE ONLY - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import anything under oracle/.

* `Binary` (reference: protocol/src/helpers.rs:174-216) serialises as `data_encoding::base64::encode` and parses with
  `data_encoding::base64::decode`: RFC 4648 section 4, standard alphabet, '=' padding, and a STRICT decoder (length a
  multiple of 4, no foreign characters, canonical trailing bits).  data-encoding 1.x is not vendored in /root/reference;
  the format is the published RFC, restated here with Python's `base64` as the independent implementation and pinned
  by the RFC 4648 section 10 test vectors (tests/test_oracle.py).
* SDAJOBv1 has no reference counterpart (the reference ships the job as JSON, resources.rs:128-139); this is an
  independent restatement of the layout in include/sda_hip.h so that the C functions are checked against something
  they do not share code with.
"""
from __future__ import annotations

import base64
import binascii
import struct
from typing import List, Sequence, Tuple

MAGIC = b"SDAJOBv1"
HEADER = 64
SEALED, VARINT, BASE64_TEXT = 0, 1, 2

RFC4648_VECTORS = [(b"", b""), (b"f", b"Zg=="), (b"fo", b"Zm8="), (b"foo", b"Zm9v"), (b"foob", b"Zm9vYg=="),
                   (b"fooba", b"Zm9vYmE="), (b"foobar", b"Zm9vYmFy")]          # RFC 4648 section 10


def binary_to_base64(raw: bytes) -> bytes:
    """Binary::to_base64 - helpers.rs:178-180"""
    return base64.b64encode(raw)


def binary_from_base64(text: bytes) -> bytes:
    """Binary::from_base64 - helpers.rs:182-184; raises ValueError("Base64 decoding error") like the reference's Err"""
    try:
        raw = base64.b64decode(text, validate=True)
    except (binascii.Error, ValueError) as e:
        raise ValueError(f"Base64 decoding error: {e}")
    if len(text) % 4 or base64.b64encode(raw) != bytes(text):      # strict: canonical padding and trailing bits
        raise ValueError("Base64 decoding error: non-canonical encoding")
    return raw


def slot_size(max_payload: int) -> int:
    return (max_payload + 15) // 16 * 16


def build_job(kind: int, payloads: Sequence[bytes], slot: int | None = None) -> bytes:
    rows = len(payloads)
    slot = slot_size(max((len(p) for p in payloads), default=0)) if slot is None else slot
    assert slot % 16 == 0 and all(len(p) <= slot for p in payloads)
    lengths_off = HEADER
    payload_off = (HEADER + 8 * rows + 15) // 16 * 16
    total = payload_off + rows * slot
    out = bytearray(total)
    out[:HEADER] = MAGIC + struct.pack("<IIQQQQQQ", HEADER, kind, rows, slot, lengths_off, payload_off, total, 0)
    for r, p in enumerate(payloads):
        struct.pack_into("<Q", out, lengths_off + 8 * r, len(p))
        out[payload_off + r * slot:payload_off + r * slot + len(p)] = p
    return bytes(out)


def parse_job(blob: bytes) -> Tuple[int, int, List[bytes]]:
    if len(blob) < HEADER or blob[:8] != MAGIC:
        raise ValueError("not an SDAJOBv1 container")
    hb, kind, rows, slot, lengths_off, payload_off, total, reserved = struct.unpack_from("<IIQQQQQQ", blob, 8)
    if (hb != HEADER or kind > BASE64_TEXT or slot % 16 or lengths_off != HEADER or reserved != 0
            or payload_off != (HEADER + 8 * rows + 15) // 16 * 16 or total != payload_off + rows * slot or len(blob) < total):
        raise ValueError("inconsistent SDAJOBv1 header")
    out = []
    for r in range(rows):
        (n,) = struct.unpack_from("<Q", blob, lengths_off + 8 * r)
        if n > slot:
            raise ValueError("row longer than its slot")
        out.append(bytes(blob[payload_off + r * slot:payload_off + r * slot + n]))
    return kind, slot, out
