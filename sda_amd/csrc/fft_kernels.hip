// Packed-Shamir share generation in tss's own form (packed_shamir.rs:42 -> tss::packed::PackedSecretSharing::share,
// SURVEY.md App. B): values [0, secrets, draws] -> radix-2 inverse transform over the k+t+1 secret nodes ->
// zero-extension -> radix-3 forward transform over the n+1 share points; shares = evaluations 1..n.
//
// This is the path for LARGE tss-valid shapes (k + t > 32, e.g. tss's PSS_155_728_100: k=100, t=155, n=728), where the
// dense n x (k+t) matrix form costs n (k+t) multiply-accumulates per batch (185,640) against ~2,200 butterflies here.
// One workgroup owns a group of G batches (8 = the batches one CSPRNG block serves, or 1 when a batch alone fills the
// LDS); values live in LDS as lazily reduced signed 64-bit numbers in [-p, p); every multiplication is ONE balanced
// 31-bit-limb Montgomery product by a table constant (exactness and register bounds: tests/test_fft_model.py).
// VALU-bound by the butterfly arithmetic; HBM traffic is the algorithmic 8 B in + 8 n / k B out per element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "chacha.hpp"
#include "kernels.hpp"
#include "modarith.hpp"

namespace sda {

__device__ __forceinline__ int32_t f_sext31(uint32_t x) { return ((int32_t)(x << 1)) >> 1; }

// x in [-p, p] times the packed constant c (lo32 = m0, hi32 = m1; Montgomery form, R = 2^62): result in (-p, p)
__device__ __forceinline__ int64_t f_mulc(int64_t x, uint64_t c, const L31Params& P) {
    const int32_t m0 = (int32_t)(uint32_t)c, m1 = (int32_t)(uint32_t)(c >> 32);
    const int32_t x0 = f_sext31((uint32_t)x);
    const int32_t x1 = (int32_t)((x - x0) >> 31);
    int64_t C0 = (int64_t)m0 * x0;
    const int64_t C1 = (int64_t)m0 * x1 + (int64_t)m1 * x0;
    const int64_t C2 = (int64_t)m1 * x1;
    const int32_t q0 = f_sext31((uint32_t)C0 * P.pinvB);
    C0 += (int64_t)P.p0 * q0;
    int64_t E = (int64_t)P.p1 * q0 + (C0 >> 31);
    const int32_t q1 = f_sext31(((uint32_t)C1 + (uint32_t)E) * P.pinvB);
    E += (int64_t)P.p0 * q1;
    return (int64_t)P.p1 * q1 + C2 + (C1 >> 31) + ((E + 0x7FFFFFFF) >> 31);
}
// [-2p, 2p) -> [-p, p)
__device__ __forceinline__ int64_t f_narrow(int64_t x, int64_t p) { return x >= 0 ? x - p : x + p; }
// canonical residue -> centred (-p/2, p/2]
__device__ __forceinline__ int64_t f_centre(uint64_t v, const L31Params& P) { return (int64_t)(v >= P.h ? v - P.p : v); }

__device__ __forceinline__ uint32_t f_bitrev(uint32_t i, uint32_t bits) { return bits ? __brev(i) >> (32 - bits) : 0u; }
__device__ __forceinline__ uint32_t f_trirev(uint32_t i, uint32_t digits) {
    uint32_t r = 0;
    for (uint32_t d = 0; d < digits; ++d) {
        const uint32_t q = i / 3u;
        r = r * 3u + (i - 3u * q);
        i = q;
    }
    return r;
}

// one uniform value per (stream, batch, draw) - sda-drbg-v1, identical to drbg_pair() of sda_kernels.hip
template <int ROUNDS>
__device__ __noinline__ uint64_t f_drbg_retry(const DrbgKey& key, uint64_t stream, uint64_t I, const ModParams& mod) {
    const uint32_t k[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
    uint64_t val = 0;
    for (uint32_t a = 1; a < 256; ++a) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(k, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (a << 24), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t x = ((uint64_t)o[2 * j] << 32) | o[2 * j + 1];
            if (lemire_sample(x, mod.m, mod.lemire_thr, val)) return val;
        }
    }
    return val;
}

template <int ROUNDS>
__global__ void packed_gen_fft_kernel(GenLayout L, ModParams mod, L31Params lp, DrbgKey key, FftPlan F, uint64_t groups,
                                      uint64_t batches) {
    extern __shared__ int64_t lds[];
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    const uint64_t p = blockIdx.x / groups, g = blockIdx.x - p * groups;
    const uint32_t G = F.G, m2 = F.m2, m3 = F.m3, k = F.k, t = F.t;
    int64_t* X = lds;                         // [G][m2]  secret-node values / coefficients
    int64_t* Y = lds + (size_t)G * m2;        // [G][m3]  share-point values
    const int64_t P = (int64_t)lp.p;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    const uint64_t stream = L.first_participant + p;
    const uint64_t b_first = g * G;

    // ---- values: [0, secrets (zero-padded, batched.rs:37-43), draws] ------------------------------------------
    for (uint32_t u = tid; u < G * (k + 1); u += T) {
        const uint32_t j = u / (k + 1), i = u - j * (k + 1);
        const uint64_t b = b_first + j;
        int64_t v = 0;
        if (i > 0) {
            const uint64_t e = b * k + (i - 1);
            if (b < batches && e < L.len) v = f_centre(canon_i64(sp[e], mod.m, mod.mu), lp);
        }
        X[(size_t)j * m2 + i] = v;
    }
    if (rp) {
        for (uint32_t u = tid; u < G * t; u += T) {
            const uint32_t j = u / t, i = u - j * t;
            const uint64_t b = b_first + j;
            X[(size_t)j * m2 + 1 + k + i] = b < batches ? f_centre(canon_i64(rp[b * t + i], mod.m, mod.mu), lp) : 0;
        }
    } else if (G == 8) {                      // one CSPRNG block serves draw i of the 8 batches of this group
        const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
        for (uint32_t i = tid; i < t; i += T) {
            const uint64_t I = g * (uint64_t)t + i;
            uint32_t o[16];
            chacha_block_lane<ROUNDS>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = jj >> 1, e = jj & 1;
                const uint64_t xw = ((uint64_t)o[8 * e + c] << 32) | o[8 * e + 4 + c];
                uint64_t val;
                if (!lemire_sample(xw, mod.m, mod.lemire_thr, val))
                    val = f_drbg_retry<ROUNDS>(key, stream, (b_first + jj) * (uint64_t)t + i, mod);
                X[(size_t)jj * m2 + 1 + k + i] = f_centre(val, lp);
            }
        }
    } else {                                  // G == 1: this batch uses its two words of each block
        const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
        const uint64_t b = b_first;
        const int c = (int)((b & 7) >> 1), e = (int)(b & 1);
        for (uint32_t i = tid; i < t; i += T) {
            const uint64_t I = (b >> 3) * (uint64_t)t + i;
            uint32_t o[16];
            chacha_block_lane<ROUNDS>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
            uint32_t hi = 0, lo = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                if (w == 8 * e + c) hi = o[w];
                if (w == 8 * e + 4 + c) lo = o[w];
            }
            uint64_t val;
            if (!lemire_sample(((uint64_t)hi << 32) | lo, mod.m, mod.lemire_thr, val))
                val = f_drbg_retry<ROUNDS>(key, stream, b * (uint64_t)t + i, mod);
            X[1 + k + i] = f_centre(val, lp);
        }
    }
    __syncthreads();

    // ---- radix-2 inverse transform, decimation in frequency: natural order in, bit-reversed order out ------------
    const uint32_t half2 = m2 >> 1;
    for (uint32_t m = m2, lg = F.a; m >= 2; m >>= 1, --lg) {
        const uint32_t h = m >> 1, step = m2 / m;
        for (uint32_t u = tid; u < G * half2; u += T) {
            const uint32_t j = u >> (F.a - 1), q = u & (half2 - 1);
            const uint32_t blk = q >> (lg - 1), jj = q & (h - 1);
            int64_t* x = X + (size_t)j * m2 + (size_t)blk * m + jj;
            const int64_t a = x[0], b = x[h];
            x[0] = f_narrow(a + b, P);
            const int64_t d = f_narrow(a - b, P);
            x[h] = jj ? f_mulc(d, F.tw2[jj * step], lp) : d;
        }
        __syncthreads();
    }

    // ---- zero-extension: coefficient j, scaled by 1 / m2, to its digit-reversed position ---------------------------
    for (uint32_t u = tid; u < G * m3; u += T) Y[u] = 0;
    __syncthreads();
    for (uint32_t u = tid; u < G * m2; u += T) {
        const uint32_t j = u >> F.a, i = u & (m2 - 1);
        Y[(size_t)j * m3 + f_trirev(i, F.b)] = f_mulc(X[(size_t)j * m2 + f_bitrev(i, F.a)], F.scale, lp);
    }
    __syncthreads();

    // ---- radix-3 forward transform, decimation in time: digit-reversed order in, natural order out ----------------
    const uint32_t third = m3 / 3;
    const uint32_t magic_third = third > 1 ? (uint32_t)(0x100000000ull / third) + 1u : 0u;   // exact for u * third < 2^32
    uint32_t t3 = 1;
    for (uint32_t s = 0; s < F.b; ++s, t3 *= 3) {
        const uint32_t m = 3 * t3, step = m3 / m;
        const uint32_t magic = t3 > 1 ? (uint32_t)(0x100000000ull / t3) + 1u : 0u;       // exact division of q < 2^16 by t3
        for (uint32_t u = tid; u < G * third; u += T) {
            const uint32_t j = third > 1 ? __umulhi(u, magic_third) : u, q = u - j * third;
            const uint32_t blk = t3 > 1 ? __umulhi(q, magic) : q, jj = q - blk * t3;
            int64_t* y = Y + (size_t)j * m3 + (size_t)blk * m + jj;
            const int64_t A = y[0];
            int64_t Bv = y[t3], C = y[2 * t3];
            if (jj) {
                Bv = f_mulc(Bv, F.tw3[jj * step], lp);
                C = f_mulc(C, F.tw3[2 * jj * step], lp);
            }
            const int64_t w = f_mulc(f_narrow(Bv - C, P), F.omega, lp);
            y[0] = f_narrow(f_narrow(A + Bv, P) + C, P);              // three-term sums in two steps: 3p does not fit 64 bits
            y[t3] = f_narrow(f_narrow(A - C, P) + w, P);
            y[2 * t3] = f_narrow(f_narrow(A - Bv, P) - w, P);
        }
        __syncthreads();
    }

    // ---- shares = evaluations 1..n, canonical, clerk-major (batched.rs:46-48) -------------------------------------
    int64_t* op = L.out + p * L.out_stride_participant;
    for (uint32_t u = tid; u < G * F.n; u += T) {
        const uint32_t sj = G == 8 ? u >> 3 : u, jj = G == 8 ? u & 7u : 0u;   // batch fastest: G consecutive values per clerk row
        const uint64_t b = b_first + jj;
        if (b >= batches) continue;
        const int64_t v = Y[(size_t)jj * m3 + sj + 1];
        op[(size_t)sj * L.out_stride_clerk + b] = v < 0 ? v + P : v;
    }
}

hipError_t launch_packed_generate_fft(const GenLayout& L, const ModParams& mod, const L31Params& lp, const DrbgKey& key,
                                      const FftPlan& F, int rounds, hipStream_t s) {
    const uint64_t batches = (L.len + F.k - 1) / F.k;
    const uint64_t groups = (batches + F.G - 1) / F.G;
    if (groups * L.participants == 0) return hipSuccess;
    const size_t lds = (size_t)F.G * ((size_t)F.m2 + F.m3) * 8;
    // 512 threads = 4 waves per SIMD with the two workgroups a CU's LDS holds (PSS_155_728_100: 58 ms per 500 x 1 Mi tile
    // against 66 with 256 threads, 79 with 1024, 120 with 128); a zero-input shortcut in the first radix-3 level, where
    // two of three inputs are zero-extension zeros, was measured 7 % SLOWER (divergence) and dropped
    const unsigned threads = F.G == 1 && F.m3 > 2187 ? 1024u : 512u;
    auto kern = rounds == 20 ? packed_gen_fft_kernel<20> : rounds == 12 ? packed_gen_fft_kernel<12> : packed_gen_fft_kernel<8>;
    if (rounds != 20 && rounds != 12 && rounds != 8) return hipErrorInvalidValue;
    if (lds > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    const uint64_t max_blocks = 0x7FFFFFFFull;
    uint64_t per = max_blocks / groups;
    if (per == 0) return hipErrorInvalidConfiguration;
    if (per > L.participants) per = L.participants;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        GenLayout S = L;
        const uint64_t cnt = per < L.participants - p0 ? per : L.participants - p0;
        S.secrets = L.secrets + p0 * L.secrets_stride;
        if (L.rand) S.rand = L.rand + p0 * L.rand_stride;
        S.out = L.out + p0 * L.out_stride_participant;
        S.participants = cnt;
        S.first_participant = L.first_participant + p0;
        kern<<<dim3((unsigned)(groups * cnt)), dim3(threads), lds, s>>>(S, mod, lp, key, F, groups, batches);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

size_t fft_lds_bytes(uint32_t m2, uint32_t m3, uint32_t G) { return (size_t)G * ((size_t)m2 + m3) * 8; }

}  // namespace sda
