#!/usr/bin/env python3
"""RFC 4648 base64 of `Binary` payloads on the device (protocol/src/helpers.rs:174-216): encode and
strict decode of a clerk job's sealed boxes (2000 rows of 3.14 MB, the config-3 tile), HIP-event timed.  GPU box."""
import ctypes as C
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBytes, synchronize  # noqa: E402

lib = capi.load()
rows = int(os.environ.get("ROWS", "2000"))
n = 3_143_078                                       # bytes of one sealed share vector of config 3 (349,526 values + 48)
in_slot = (n + 15) // 16 * 16
tlen = lib.sda_base64_encoded_size(n)
text_slot = (tlen + 15) // 16 * 16
rng = np.random.default_rng(1)
row = rng.integers(0, 256, size=in_slot, dtype=np.uint8).tobytes()
d_in = DeviceBytes(rows * in_slot)
for r in range(0, rows, 50):                         # fill by chunks (the content does not matter to the kernels)
    capi.check(lib.sda_dev_upload(d_in.ptr + r * in_slot, row * min(50, rows - r), in_slot * min(50, rows - r)))
d_len = DeviceBytes.from_bytes(np.full(rows, n, dtype="<u8").tobytes())
d_text, d_tlen = DeviceBytes(rows * text_slot), DeviceBytes(rows * 8)
d_out, d_olen = DeviceBytes(rows * in_slot), DeviceBytes(rows * 8)
d_status = DeviceBytes(4).zero()


def timed(fn, reps=5):
    a, b = C.c_void_p(), C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(a))); capi.check(lib.sda_event_create(C.byref(b)))
    fn(); synchronize()
    capi.check(lib.sda_event_record(a, None))
    for _ in range(reps):
        fn()
    capi.check(lib.sda_event_record(b, None))
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(a, b, C.byref(ms)))
    return ms.value / reps


enc = timed(lambda: crypto.base64_encode_rows_dev(d_in.ptr, in_slot, d_len.ptr, rows, n, d_text.ptr, text_slot, d_tlen.ptr))
dec = timed(lambda: crypto.base64_decode_rows_dev(d_text.ptr, text_slot, d_tlen.ptr, rows, tlen, d_out.ptr, in_slot, d_olen.ptr, d_status.ptr))
ok = (d_status.to_bytes() == bytes(4) and d_out.to_bytes(n, 0) == row[:n] and
      np.frombuffer(d_olen.to_bytes(), dtype="<u8").tolist() == [n] * rows)
raw, text = rows * n, rows * tlen
print(json.dumps({"job": f"{rows} payloads of {n} bytes <-> {tlen} characters", "encode_ms": enc, "decode_ms": dec,
                  "encode_GBps_of_traffic": (raw + text) / (enc * 1e-3) / 1e9, "decode_GBps_of_traffic": (raw + text) / (dec * 1e-3) / 1e9,
                  "payload_GBps_encode": raw / (enc * 1e-3) / 1e9, "payload_GBps_decode": raw / (dec * 1e-3) / 1e9,
                  "round_trip_verified": ok}, indent=1))
