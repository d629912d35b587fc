// helpers shared by the translation units of the C ABI (sda_capi.cpp owns them)
#pragma once
#include <stdint.h>

#include "kernels.hpp"

int capi_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // sets sda_last_error(), returns code
int capi_make_mod(int64_t modulus, sda::ModParams& mod);                                  // validated Barrett / Lemire constants
int capi_device_ready();                                                                  // a device exists; make the selected one current

// ---- path-selection knobs (A/B measurements and parity tests of the non-default kernels) -----------------------------------
// A release build of the library reads NO environment variable: a knob changes only through the test-only entry point
// sda_debug_set_knob (include/sda_hip_debug.h).  Built with -DSDA_AB_KNOBS (tools/build_ab_variant.sh, never build()), a
// knob left unset falls back to the environment variable of the same name, for shell-driven A/B runs.
namespace sda {
enum Knob {
    KNOB_FORCE_GENERIC, KNOB_FORCE_MONT64, KNOB_FORCE_FFT, KNOB_FORCE_MFMA, KNOB_NO_MFMA, KNOB_NO_SIDE_STREAM,
    KNOB_SIDE_STREAM_WGS, KNOB_SIDE_STREAM_PRIORITY_HIGH, KNOB_FFT_G, KNOB_FFT_THREADS, KNOB_VARINT_PATH /* 1 stream, 2 scan */,
    KNOB_FORCE_COLLECTIVES, KNOB_NO_NARROW, KNOB_WIRE_WG_PER_CU, KNOB_SBOX_WG_PER_CU, KNOB_NO_LAZY, KNOB_NO_XCD_MAP, KNOB_NO_NGEMM, KNOB_NO_WIDE_GROUP, KNOB_NGEMM_CLERK_WG, KNOB_NO_KARATSUBA, KNOB_COUNT
};
long knob(Knob k);             // 0 = unset / default
// an UNUSED dynamic-LDS request that caps the resident workgroups of a launch at wg_per_cu per CU (0: no cap), so that a
// kernel bound by one resource leaves wave slots to a kernel bound by another on a second stream (DESIGN.md 5 "Schedules")
inline unsigned residency_pad_bytes(long wg_per_cu, unsigned static_lds) {
    if (wg_per_cu <= 0) return 0u;
    const unsigned share = (160u * 1024u) / (unsigned)wg_per_cu;
    const unsigned pad = share > static_lds + 256u ? share - static_lds - 256u : 0u;
    return pad > 64u * 1024u - static_lds ? 64u * 1024u - static_lds : pad;        // within the default per-workgroup limit
}
}  // namespace sda
