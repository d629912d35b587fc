#!/usr/bin/env python3
"""Wire-format rows: encode, decode (both forms), decode -> clerk-sum, on tiles of
`rows` encoded vectors of L = 349526 canonical 62-bit shares.  Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402
from sda_amd.device import DeviceBuffer, synchronize  # noqa: E402

P62 = 4611686006577364993
lib = capi.load()


def timed(fn, reps=5):
    fn()
    synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


out = {}
L, stride = 349526, 349536
for rows in [int(x) for x in os.environ.get("ROWS", "2000,16000").split(",")]:
    vals = DeviceBuffer(rows * stride)
    capi.check(lib.sda_fill_synthetic_dev(vals.ptr, rows, stride, stride, 0, 9, P62, None))
    codec = crypto.VarintCodec()
    cap = rows * L * 10
    d_bytes = DeviceBuffer((cap + 7) // 8)
    d_off = DeviceBuffer(rows + 1)
    total = [0]

    def enc():
        total[0] = codec.encode_dev(vals.ptr, rows, L, stride, d_bytes.ptr, cap, d_off.ptr)
    dt = timed(enc, reps=3)
    nv = rows * L
    out[f"encode_{rows}x{L}"] = {"ms": dt * 1e3, "values_per_s": nv / dt, "wire_bytes": total[0],
                                 "GBps_algorithmic": (nv * 8 + total[0]) / dt / 1e9}
    dec = DeviceBuffer(rows * stride)
    st = DeviceBuffer(1).zero()
    comb = crypto.ShareCombiner(crypto.Additive(3, P62))
    o2 = DeviceBuffer(L)
    ref = None
    capi.use_test_hooks()                                        # both decode forms: the knob table exists in libsda_hip_test.so only
    for path in ("scan", "stream"):
        capi.check(capi.load().sda_debug_set_knob(b"SDA_VARINT_PATH", 1 if path == "stream" else 2))
        dt = timed(lambda: codec.decode_dev(d_bytes.ptr, total[0], d_off.ptr, rows, L, dec.ptr, stride, st.ptr), reps=3)
        assert st.to_numpy()[0] == 0

        def dec_comb():
            comb.begin_dev(1, L)
            comb.update_encoded_dev(codec, d_bytes.ptr, total[0], d_off.ptr, rows, st.ptr)
            comb.finish_dev(o2.ptr)
        dt2 = timed(dec_comb, reps=3)
        assert st.to_numpy()[0] == 0
        got = o2.to_numpy()
        if ref is None:
            ref = got
        same = bool((got == ref).all())
        out[f"decode_{path}_{rows}x{L}"] = {"ms": dt * 1e3, "values_per_s": nv / dt,
                                           "GBps_algorithmic": (nv * 8 + total[0]) / dt / 1e9}
        out[f"wire_clerk_sum_{path}_{rows}x{L}"] = {"ms": dt2 * 1e3, "values_per_s": nv / dt2,
                                                   "GBps_wire_bytes": total[0] / dt2 / 1e9, "sums_equal_first_form": same}
    capi.use_release()
    # slotted rows: single pass on both sides
    del d_bytes
    slot = codec.slot_size(L)
    d_slots = DeviceBuffer(rows * slot // 8 + 2)
    d_len = DeviceBuffer(rows)
    dt = timed(lambda: codec.encode_rows_dev(vals.ptr, rows, L, stride, d_slots.ptr, slot, d_len.ptr), reps=3)
    wire = int(d_len.to_numpy().sum())
    assert wire == total[0]
    out[f"encode_slotted_{rows}x{L}"] = {"ms": dt * 1e3, "values_per_s": nv / dt, "wire_bytes": wire,
                                         "GBps_algorithmic": (nv * 8 + wire) / dt / 1e9}
    dt = timed(lambda: codec.decode_rows_dev(d_slots.ptr, slot, d_len.ptr, rows, L, dec.ptr, stride, st.ptr), reps=3)
    assert st.to_numpy()[0] == 0
    out[f"decode_slotted_{rows}x{L}"] = {"ms": dt * 1e3, "values_per_s": nv / dt, "GBps_algorithmic": (nv * 8 + wire) / dt / 1e9}

    def rows_comb():
        comb.begin_dev(1, L)
        comb.update_encoded_rows_dev(codec, d_slots.ptr, slot, d_len.ptr, rows, st.ptr)
        comb.finish_dev(o2.ptr)
    dt = timed(rows_comb, reps=3)
    assert st.to_numpy()[0] == 0
    out[f"wire_clerk_sum_slotted_{rows}x{L}"] = {"ms": dt * 1e3, "values_per_s": nv / dt, "GBps_wire_bytes": wire / dt / 1e9,
                                                 "sums_equal_first_form": bool((o2.to_numpy() == ref).all())}
    del vals, d_slots, dec
print(json.dumps(out, indent=1))
