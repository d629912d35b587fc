// The per-lane form of the sda-drbg-v1 draws shared by the transform kernel (fft_kernels.hip) and the narrow limb GEMM
// (ngemm_kernels.hip): one lane computes a whole ChaCha block = draw i of 8 consecutive batches.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "chacha.hpp"
#include "kernels.hpp"
#include "modarith.hpp"

namespace sda {

// Lemire sampling (sda-drbg-v1: accept iff lo64(x m) >= 2^64 mod m, value hi64(x m)) for a modulus below 2^32: the 96-bit
// product is two 32 x 32 -> 64 multiply-adds instead of a 64 x 64 one
__device__ __forceinline__ bool f_lemire32(uint64_t x, uint32_t m, uint64_t thr, uint64_t& out) {
    const uint64_t t0 = (uint64_t)(uint32_t)x * m;
    const uint64_t t1 = (uint64_t)(uint32_t)(x >> 32) * m + (t0 >> 32);
    out = t1 >> 32;
    return ((t1 << 32) | (uint32_t)t0) >= thr;
}
template <typename V> __device__ __forceinline__ bool f_lemire(uint64_t x, const ModParams& mod, uint64_t& out) {
    if constexpr (sizeof(V) == 4) return f_lemire32(x, (uint32_t)mod.m, mod.lemire_thr, out);
    else return lemire_sample(x, mod.m, mod.lemire_thr, out);
}

// one uniform value per (stream, batch, draw) - sda-drbg-v1, identical to drbg_pair() of sda_kernels.hip.  The key travels
// BY VALUE: a by-reference key in this rare path makes every lane park the key in scratch memory at kernel entry
template <int ROUNDS>
__device__ __noinline__ uint64_t f_drbg_retry(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5, uint32_t k6,
                                              uint32_t k7, uint64_t stream, uint64_t I, uint64_t m, uint64_t lemire_thr) {
    const uint32_t k[8] = {k0, k1, k2, k3, k4, k5, k6, k7};
    uint64_t val = 0;
    for (uint32_t a = 1; a < 256; ++a) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(k, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (a << 24), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t x = ((uint64_t)o[2 * j] << 32) | o[2 * j + 1];
            if (lemire_sample(x, m, lemire_thr, val)) return val;
        }
    }
    return val;
}

// the paired rule's retry stream (modarith.hpp): draw pair j of batch b, counter I = b * ceil(T / 2) + j
template <int ROUNDS>
__device__ __noinline__ uint64_t f_drbg_retry_pair(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5, uint32_t k6,
                                                   uint32_t k7, uint64_t stream, uint64_t I, uint64_t m, uint64_t thr2) {
    const uint32_t k[8] = {k0, k1, k2, k3, k4, k5, k6, k7};
    uint32_t ra = 0, rb = 0;
    for (uint32_t a = 1; a < 256; ++a) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(k, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (a << 24), o);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lemire_pair(((uint64_t)o[2 * j] << 32) | o[2 * j + 1], (uint32_t)m, thr2, ra, rb)) return ((uint64_t)rb << 32) | ra;
    }
    return ((uint64_t)rb << 32) | ra;
}
// one candidate word -> the two draws of a pair (retry stream on the rare rejection); returns rb << 32 | ra
template <int ROUNDS>
__device__ __forceinline__ uint64_t f_draw_pair(uint64_t xw, const uint32_t (&kk)[8], uint64_t stream, uint64_t retry_I, uint64_t m, uint64_t thr2) {
    uint32_t ra, rb;
    if (__builtin_expect(lemire_pair(xw, (uint32_t)m, thr2, ra, rb), 1)) return ((uint64_t)rb << 32) | ra;
    return f_drbg_retry_pair<ROUNDS>(kk[0], kk[1], kk[2], kk[3], kk[4], kk[5], kk[6], kk[7], stream, retry_I, m, thr2);
}

}  // namespace sda
