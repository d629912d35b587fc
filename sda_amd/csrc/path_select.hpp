// Which kernel family serves a packed-Shamir share generator - ONE pure host function, no device, no globals.
//
// The reference dispatches on the scheme enum only (client/src/crypto/sharing/mod.rs:35-55: Additive -> additive.rs,
// PackedShamir -> packed_shamir.rs); every family below computes the SAME shares for the same inputs, so the choice is a
// performance decision and nothing else.  It is made here and nowhere else: sda_share_generator_new() builds what
// select_path() names, the calls dispatch through path_for_call() / fused_for_call(), bench.py prints what the library
// reports (sda_share_generator_path_name, sda_debug_last_kernel) and tests/test_path_select.py pins the table below
// through sda_debug_select_path() on a machine without a GPU.
//
//   shape / modulus                                                        -> family
//   additive                                                               -> additive
//   p < 2^31, k + t <= 16, matrix fits the kernarg segment                 -> n31     (one 32-bit limb per residue)
//   p <= 0x7F7F7F (three base-256 digits), k + t > 16                      -> ngemm   (limb GEMM, v_mfma_i32_16x16x64_i8)
//   tss-valid transform shape (k+t+1 = 2^a, n+1 = 3^b, roots of those orders), k + t > 32  -> fft (tss's own algorithm)
//   k + t in 12..16, not a three-digit limb-31 shape, n <= the MFMA table  -> mfma    (62-bit limb GEMM)
//   compiled (k, t) or k + t <= 16, n (k + t) entries fit the kernarg      -> l31     (balanced 31-bit limbs; R = 2^62 or 2^93)
//   compiled 64-bit Montgomery shape (superseded; knob SDA_FORCE_MONT64)   -> mont64
//   k + t <= 64                                                            -> l31_global (matrix in global memory)
//   anything else                                                          -> generic (draws materialised first)
// A narrow family (n31 / ngemm) is an OVERLAY: the wide family chosen for the shape stays built and serves the calls the
// narrow kernels do not (ChaCha12 / ChaCha8 A/B runs draw through the wide kernels).
#pragma once
#include <stdint.h>

#include "kernels.hpp"

namespace sda {

enum WidePath { WIDE_GENERIC = 0, WIDE_MONT64, WIDE_L31, WIDE_L31_GLOBAL, WIDE_MFMA, WIDE_FFT };
enum NarrowPath { NARROW_NONE = 0, NARROW_N31, NARROW_NGEMM };
// the family a CALL runs (what rocprof shows, modulo template arguments)
enum GenFamily { FAM_ADDITIVE = 0, FAM_ADDITIVE_SIGNED, FAM_N31, FAM_NGEMM, FAM_L31, FAM_MONT64, FAM_L31_GLOBAL, FAM_FFT, FAM_MFMA, FAM_GENERIC };

// knobs that take part in the selection (include/sda_hip_debug.h), snapshotted into the handle when it is created
struct PathKnobs {
    bool force_generic = false, force_mont64 = false, force_fft = false, force_mfma = false;
    bool no_mfma = false, no_narrow = false, no_ngemm = false;
};

// facts about the scheme that need its constants (host arithmetic only, sda_capi.cpp computes them)
struct PathFacts {
    bool transform_shape = false;   // k + t + 1 = 2^a = ord(omega_secrets), n + 1 = 3^b = ord(omega_shares), the group fits LDS
    bool eight_term_ok = true;      // three-digit limb-31 shapes with k + t = 1 mod 7: the 8-term last group passes on the constants
};

struct PathChoice {
    WidePath wide = WIDE_GENERIC;
    NarrowPath narrow = NARROW_NONE;
    unsigned l31_r_bits = 62;       // Montgomery radix of the limb-31 constants (62, or 93 for the compiled three-digit shapes)
};

inline PathChoice select_path(uint32_t k, uint32_t t, uint32_t n, uint64_t p, const PathFacts& f, const PathKnobs& kn) {
    PathChoice c;
    const uint32_t kt = k + t;
    const bool wide_forced = kn.force_generic || kn.force_mont64;
    bool l31 = packed_l31_path_available(k, t, n) && !wide_forced;
    const unsigned rb = packed_l31_r_bits(k, t);
    if (l31 && rb == 93 && !f.eight_term_ok) l31 = false;
    const bool mont64 = !l31 && packed_fast_path_available(k, t, n) && !kn.force_generic;
    const bool l31g = !l31 && !mont64 && packed_l31_global_path_available(k, t) && !wide_forced;
    const bool narrow_ok = !kn.no_narrow;
    if (!wide_forced && (kt > 32 || kn.force_fft) && f.transform_shape) {
        c.wide = WIDE_FFT;
        // below 2^23 the dense product on the matrix cores is ahead of the transform (the narrow transform kernel itself needs p < 2^30)
        if (narrow_ok && p < (1ull << 30) && !kn.no_ngemm && kt > 16 && packed_ngemm_path_available(k, t, p)) c.narrow = NARROW_NGEMM;
        return c;
    }
    if (packed_mfma_path_available(k, t, n) && !wide_forced && !kn.no_mfma && ((kt >= 12 && !(l31 && rb == 93)) || kn.force_mfma))
        c.wide = WIDE_MFMA;
    else if (l31) { c.wide = WIDE_L31; c.l31_r_bits = rb; }
    else if (mont64) c.wide = WIDE_MONT64;
    else if (l31g) c.wide = WIDE_L31_GLOBAL;
    else c.wide = WIDE_GENERIC;
    const bool forced = wide_forced || kn.force_mfma || kn.force_fft;
    if (!forced && narrow_ok) {
        if (packed_n31_path_available(k, t, n, p)) c.narrow = NARROW_N31;
        else if (!kn.no_ngemm && kt > 16 && packed_ngemm_path_available(k, t, p)) c.narrow = NARROW_NGEMM;
    }
    return c;
}

// the narrow kernels draw with ChaCha20 only; injected randomness draws nothing
inline bool narrow_serves(bool injected_rand, int rounds) { return injected_rand || rounds == 20; }

// separate share-generation launch (generate / generate_batch_dev)
inline GenFamily path_for_call(const PathChoice& c, bool injected_rand, int rounds) {
    if (c.narrow == NARROW_N31 && narrow_serves(injected_rand, rounds)) return FAM_N31;
    if (c.narrow == NARROW_NGEMM && narrow_serves(injected_rand, rounds)) return FAM_NGEMM;
    switch (c.wide) {
        case WIDE_L31: return FAM_L31;
        case WIDE_MONT64: return FAM_MONT64;
        case WIDE_L31_GLOBAL: return FAM_L31_GLOBAL;
        case WIDE_FFT: return FAM_FFT;
        case WIDE_MFMA: return FAM_MFMA;
        default: return FAM_GENERIC;
    }
}

// dual-role launch (generate_combine_dev, always the device CSPRNG): the family with a dual-role kernel, or FAM_GENERIC for
// "none: two launches"
inline GenFamily fused_for_call(const PathChoice& c, int rounds) {
    if (c.narrow == NARROW_N31 && rounds == 20) return FAM_N31;
    if (c.narrow == NARROW_NGEMM && rounds == 20) return FAM_NGEMM;
    if (c.wide == WIDE_L31) return FAM_L31;
    if (c.wide == WIDE_MFMA) return FAM_MFMA;
    return FAM_GENERIC;
}

inline const char* family_name(GenFamily f) {
    switch (f) {
        case FAM_ADDITIVE: return "additive";
        case FAM_ADDITIVE_SIGNED: return "additive_signed";
        case FAM_N31: return "n31";
        case FAM_NGEMM: return "ngemm";
        case FAM_L31: return "l31";
        case FAM_MONT64: return "mont64";
        case FAM_L31_GLOBAL: return "l31_global";
        case FAM_FFT: return "fft";
        case FAM_MFMA: return "mfma";
        default: return "generic";
    }
}
inline const char* wide_name(WidePath w) {
    switch (w) {
        case WIDE_MONT64: return "mont64";
        case WIDE_L31: return "l31";
        case WIDE_L31_GLOBAL: return "l31_global";
        case WIDE_MFMA: return "mfma";
        case WIDE_FFT: return "fft";
        default: return "generic";
    }
}
inline const char* narrow_name(NarrowPath n) { return n == NARROW_N31 ? "n31" : n == NARROW_NGEMM ? "ngemm" : "none"; }

}  // namespace sda
