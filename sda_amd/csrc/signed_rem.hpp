// Rust's `%` on i64 (truncated remainder, sign of the dividend) for a sum or difference formed exactly in 128 bits -
// shared by the reference-representative kernels (signed_kernels.hip, the CSPRNG form in sda_kernels.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace sda {

// Rust's `x % q` for q > 0 and x = a + b (or a - b) formed exactly
__device__ __forceinline__ int64_t trunc_rem128(__int128 x, int64_t q) {
    if (x > -(__int128)q && x < (__int128)q) return (int64_t)x;                 // the common cases first: |x| < 2q
    if (x >= q && x < 2 * (__int128)q) return (int64_t)(x - q);
    if (x <= -(__int128)q && x > -2 * (__int128)q) return (int64_t)(x + q);
    return (int64_t)(x % q);                                                     // C's % truncates like Rust's
}

}  // namespace sda
