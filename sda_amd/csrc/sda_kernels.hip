// gfx950 (MI355X / CDNA4) kernels of the SDA secure-aggregation hot path.
//
// All of this is HBM-bound integer work: no MFMA, no GEMM reshaping.  The rules that
// matter are the streaming ones - 16-byte coalesced loads/stores per lane, >> 256 workgroups,
// several independent loads in flight per lane, nothing re-read - plus keeping the 64-bit modular
// arithmetic cheap enough (see modarith.hpp) that the ALU stays under the memory time.
//
// Kernel            replaces (reference file:line)                               bound
// ----------------  -----------------------------------------------------------  -------------------
// additive_gen      additive.rs:32-51 via batched.rs:18-53                        HBM write (8n B/elem)
// packed_gen        packed_shamir.rs:40-43 -> tss share, via batched.rs:18-53     HBM write (8n/k B/elem)
// combine_update    combiner.rs:15-29 (also full.rs:37-52, additive.rs:55-73)     HBM read  (8 B/value)
// packed_reconstruct packed_shamir.rs:73-77 -> tss reconstruct, batched.rs:68-97  tiny
// chacha_mask_*     chacha.rs:36-39, :60-73 (rand 0.3 ChaChaRng + gen_range)      VALU (ChaCha20)
// addsub_mod        full.rs:28-31,60-63; chacha.rs:42-45,86-89                    tiny
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "chacha.hpp"
#include "clerk_sum.hpp"
#include "kernels.hpp"
#include "modarith.hpp"
#include "signed_rem.hpp"

namespace sda {

static constexpr int kThreads = 256;

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

// =================================================================================================
// sda-drbg-v1: the on-device CSPRNG used when the caller injects no randomness.
//
//   value (stream S, batch b, draw i of T):
//     main candidate : ChaCha block with words 12,13 = I = (b >> 3) * T + i, word 14 = lo32(S),
//                      word 15 = (S >> 32) & 0xFFFFFF (attempt 0).  The block is computed by the DPP
//                      quad that owns batches 8*(b>>3) .. +7; lane c = (b & 7) >> 1, e = b & 1;
//                      x = (out[8e + c] << 32) | out[8e + 4 + c].
//     acceptance     : Lemire: accept iff lo64(x * m) >= 2^64 mod m; value = hi64(x * m).
//     retry (rare)   : attempt a = 1, 2, ...: block with words 12,13 = b * T + i, word 15 |= a << 24;
//                      candidates x_j = (out[2j] << 32) | out[2j+1], j = 0..7, first accepted wins.
//   moduli m <= 0x7F7F7F (round 5, the PAIRED rule - modarith.hpp): draws 2j, 2j + 1 of a batch come from ONE candidate word
//                      (Lemire with range m^2); main block counter I = (b >> 3) * ceil(T / 2) + j, retry counter
//                      b * ceil(T / 2) + j, same lanes and words.
// =================================================================================================
// Everything by value: a by-reference key would force a scratch copy of it at kernel entry
// (measured: +11 GB of HBM writes per 2000-participant launch).
template <int ROUNDS>
__device__ __noinline__ uint64_t drbg_retry(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5,
                                            uint32_t k6, uint32_t k7, uint64_t stream, uint64_t I, uint64_t m,
                                            uint64_t lemire_thr) {
    const uint32_t k[8] = {k0, k1, k2, k3, k4, k5, k6, k7};
    uint64_t val = 0;
    for (uint32_t a = 1; a < 256; ++a) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(k, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream,
                                  ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (a << 24), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t x = ((uint64_t)o[2 * j] << 32) | o[2 * j + 1];
            if (lemire_sample(x, m, lemire_thr, val)) return val;
        }
    }
    return val;
}
// the paired rule's retry stream: draw pair j of batch b (counter I = b * ceil(T / 2) + j); returns element `which` of the pair
template <int ROUNDS>
__device__ __noinline__ uint64_t drbg_retry_pair(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5,
                                                 uint32_t k6, uint32_t k7, uint64_t stream, uint64_t I, uint64_t m, uint64_t thr2,
                                                 uint32_t which) {
    const uint32_t k[8] = {k0, k1, k2, k3, k4, k5, k6, k7};
    uint32_t ra = 0, rb = 0;
    for (uint32_t a = 1; a < 256; ++a) {
        uint32_t o[16];
        chacha_block_lane<ROUNDS>(k, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream,
                                  ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (a << 24), o);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lemire_pair(((uint64_t)o[2 * j] << 32) | o[2 * j + 1], (uint32_t)m, thr2, ra, rb)) return which ? rb : ra;
    }
    return which ? rb : ra;
}

// this lane's column of the ChaCha input state (lane c = lane id & 3 of a DPP quad): constant, key
// word c, key word 4 + c.  Computed once per lane with selects - a dynamically indexed key would be
// placed in scratch.
struct QuadCol {
    uint32_t cst, kb, kc;
};
__device__ __forceinline__ QuadCol quad_col(const DrbgKey& key) {
    const uint32_t c = threadIdx.x & 3;
    QuadCol q;
    q.cst = c == 0 ? SDA_CHACHA_C0 : c == 1 ? SDA_CHACHA_C1 : c == 2 ? SDA_CHACHA_C2 : SDA_CHACHA_C3;
    q.kb = c == 0 ? key.w[0] : c == 1 ? key.w[1] : c == 2 ? key.w[2] : key.w[3];
    q.kc = c == 0 ? key.w[4] : c == 1 ? key.w[5] : c == 2 ? key.w[6] : key.w[7];
    return q;
}

// Two uniform values per lane (for batches b0 = 2*pair and b0+1).  ALL FOUR lanes of a quad must
// be active when this is called (DPP reads its neighbours).
template <int ROUNDS>
__device__ __forceinline__ void drbg_pair(const DrbgKey& key, const QuadCol& qc, uint64_t stream, uint64_t pair,
                                          uint32_t T, uint32_t i, const ModParams& mod, uint64_t& r0, uint64_t& r1) {
    const uint32_t c = threadIdx.x & 3;
    const uint64_t g = pair >> 2;                       // batch group of 8 = one quad
    const bool paired = drbg_paired(mod.m);             // uniform: draw i is element i & 1 of draw pair i >> 1 (this form computes the
                                                        // pair's block for each of its elements; the lane forms of the large-shape
                                                        // kernels, where the draws dominate, take both from one block)
    const uint32_t T2 = (T + 1) >> 1, j = i >> 1;
    const uint64_t I = paired ? g * (uint64_t)T2 + j : g * (uint64_t)T + i;
    const uint32_t ctr = c == 0 ? (uint32_t)I : c == 1 ? (uint32_t)(I >> 32) : c == 2 ? (uint32_t)stream
                                                                              : ((uint32_t)(stream >> 32) & 0xFFFFFFu);
    uint32_t o0, o1, o2, o3;
    chacha_block_quad<ROUNDS>(qc.cst, qc.kb, qc.kc, ctr, o0, o1, o2, o3);
    const uint64_t x0 = ((uint64_t)o0 << 32) | o1;
    const uint64_t x1 = ((uint64_t)o2 << 32) | o3;
    if (paired) {
        uint32_t a0, b0, a1, b1;
        const bool ok0 = lemire_pair(x0, (uint32_t)mod.m, mod.lemire_thr2, a0, b0);
        const bool ok1 = lemire_pair(x1, (uint32_t)mod.m, mod.lemire_thr2, a1, b1);
        r0 = (i & 1u) ? b0 : a0;
        r1 = (i & 1u) ? b1 : a1;
        if (__builtin_expect(!ok0, 0))
            r0 = drbg_retry_pair<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                         (2 * pair) * (uint64_t)T2 + j, mod.m, mod.lemire_thr2, i & 1u);
        if (__builtin_expect(!ok1, 0))
            r1 = drbg_retry_pair<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                         (2 * pair + 1) * (uint64_t)T2 + j, mod.m, mod.lemire_thr2, i & 1u);
        return;
    }
    const bool ok0 = lemire_sample(x0, mod.m, mod.lemire_thr, r0);
    const bool ok1 = lemire_sample(x1, mod.m, mod.lemire_thr, r1);
    if (__builtin_expect(!ok0, 0))
        r0 = drbg_retry<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                (2 * pair) * (uint64_t)T + i, mod.m, mod.lemire_thr);
    if (__builtin_expect(!ok1, 0))
        r1 = drbg_retry<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                (2 * pair + 1) * (uint64_t)T + i, mod.m, mod.lemire_thr);
}

// ---- small load/store helpers --------------------------------------------------------------------
// Cache policies, each measured on config 3 (interleaved A/B of builds): secrets with the default policy (nt loads
// cost the dual-role launch 12 %), share stores nt (plain: -1 %; sc1 / sc0 sc1 / sc1 nt write-through forms: equal within
// noise), clerk-sum loads nt (plain: 7.8 vs 7.2 ms).
__device__ __forceinline__ ll2 load2(const int64_t* p) { return *reinterpret_cast<const ll2*>(p); }
__device__ __forceinline__ void store2(int64_t* p, uint64_t a, uint64_t b) {
    ll2 v; v.x = (long long)a; v.y = (long long)b;
    __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(p));   // shares are written once, read much later
}

// one lane's two adjacent batches of a clerk row
template <bool VEC>
__device__ __forceinline__ void store_pair(int64_t* o, uint64_t a, uint64_t b, bool in0, bool in1) {
    if (VEC && in1) store2(o, a, b);
    else {
        if (in0) o[0] = (int64_t)a;
        if (in1) o[1] = (int64_t)b;
    }
}

// work item -> (participant, chunk)
__device__ __forceinline__ void split_item(uint64_t item, uint64_t chunks, uint64_t& p, uint64_t& chunk) {
    p = item / chunks;
    chunk = item - p * chunks;
}

// =================================================================================================
// K1  additive share generation.  One lane = two adjacent elements (16-byte accesses).
//     shares 0..n-2 are the uniform draws, share n-1 = secret - sum(draws) mod q  (additive.rs:42-47)
// =================================================================================================
template <int ROUNDS, bool VEC>
__device__ __forceinline__ void additive_gen_body(const GenLayout& L, uint32_t n, const ModParams& mod,
                                                  const DrbgKey& key, uint64_t chunks, uint64_t item) {
    uint64_t p, chunk;
    split_item(item, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    const bool in0 = b0 < L.len, in1 = b0 + 1 < L.len;

    const int64_t* sp = L.secrets + p * L.secrets_stride;
    uint64_t s0 = 0, s1 = 0;
    if (VEC && in1) {
        ll2 v = load2(sp + b0);
        s0 = canon_i64(v.x, mod.m, mod.mu);
        s1 = canon_i64(v.y, mod.m, mod.mu);
    } else {
        if (in0) s0 = canon_i64(sp[b0], mod.m, mod.mu);
        if (in1) s1 = canon_i64(sp[b0 + 1], mod.m, mod.mu);
    }

    int64_t* op = L.out + p * L.out_stride_participant + b0;
    const uint32_t T = n - 1;
    const uint64_t stream = L.first_participant + p;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    const QuadCol qc = quad_col(key);
    for (uint32_t i = 0; i < T; ++i) {
        uint64_t r0 = 0, r1 = 0;
        if (rp) {
            if (in0) r0 = canon_i64(rp[b0 * T + i], mod.m, mod.mu);
            if (in1) r1 = canon_i64(rp[(b0 + 1) * T + i], mod.m, mod.mu);
        } else {
            drbg_pair<ROUNDS>(key, qc, stream, pair, T, i, mod, r0, r1);
        }
        s0 = submod(s0, r0, mod.m);
        s1 = submod(s1, r1, mod.m);
        int64_t* o = op + (size_t)i * L.out_stride_clerk;
        if (VEC && in1) store2(o, r0, r1);
        else {
            if (in0) o[0] = (int64_t)r0;
            if (in1) o[1] = (int64_t)r1;
        }
    }
    int64_t* o = op + (size_t)T * L.out_stride_clerk;
    if (VEC && in1) store2(o, s0, s1);
    else {
        if (in0) o[0] = (int64_t)s0;
        if (in1) o[1] = (int64_t)s1;
    }
}

template <int ROUNDS, bool VEC>
__global__ __launch_bounds__(kThreads) void additive_gen_kernel(GenLayout L, uint32_t n, ModParams mod,
                                                                DrbgKey key, uint64_t chunks) {
    additive_gen_body<ROUNDS, VEC>(L, n, mod, key, chunks, blockIdx.x);
}

// The reference's own representatives (SDA_VALUES_RUST_SIGNED) with the library's randomness: the SAME draws as the kernel above
// (same streams, same indexing), the secret taken as the raw i64 and the last share folded with Rust's truncated `%`
// (additive.rs:42-47) - canonical == signed modulo q, and the draws need no scratch buffer.
template <int ROUNDS>
__global__ __launch_bounds__(kThreads) void signed_additive_gen_drbg_kernel(GenLayout L, uint32_t n, ModParams mod, DrbgKey key,
                                                                            uint64_t chunks) {
    uint64_t p, chunk;
    split_item(blockIdx.x, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    const bool in0 = b0 < L.len, in1 = b0 + 1 < L.len;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t q = (int64_t)mod.m;
    int64_t a0 = in0 ? sp[b0] : 0, a1 = in1 ? sp[b0 + 1] : 0;
    int64_t* op = L.out + p * L.out_stride_participant + b0;
    const uint32_t T = n - 1;
    const uint64_t stream = L.first_participant + p;
    const QuadCol qc = quad_col(key);
    for (uint32_t i = 0; i < T; ++i) {
        uint64_t r0 = 0, r1 = 0;
        drbg_pair<ROUNDS>(key, qc, stream, pair, T, i, mod, r0, r1);       // all four lanes of a quad take part
        a0 = trunc_rem128((__int128)a0 - (int64_t)r0, q);
        a1 = trunc_rem128((__int128)a1 - (int64_t)r1, q);
        int64_t* o = op + (size_t)i * L.out_stride_clerk;
        if (in0) o[0] = (int64_t)r0;
        if (in1) o[1] = (int64_t)r1;
    }
    int64_t* o = op + (size_t)T * L.out_stride_clerk;
    if (in0) o[0] = a0;
    if (in1) o[1] = a1;
}

// =================================================================================================
// K2  packed-Shamir share generation:  shares[n] = M[n x (K+T)] * [secrets(K) ; draws(T)]  mod p.
//     One lane = two adjacent batches.  M is in Montgomery form in the kernarg segment (SGPR
//     operands); each output is an un-reduced 128-bit dot product + one REDC.
// =================================================================================================
template <int KT>
__device__ __forceinline__ uint64_t mont_dot(const uint64_t* __restrict__ row, const uint64_t (&v)[KT],
                                             uint64_t p, uint64_t pinv) {
    U128 acc{0, 0};
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        mac128(acc, row[i], v[i]);
        // every product is < p^2 <= p*2^62: four of them keep acc < p*2^64; fold after 8, 12, ...
        if (((i + 1) & 3) == 0 && (i + 1) >= 8) mont_acc_condsub(acc, p);
    }
    if (KT > 4 && (KT & 3) != 0) mont_acc_condsub(acc, p);
    return mont_redc(acc, p, pinv);
}

template <int K, int T, int ROUNDS, bool VEC>
__global__ __launch_bounds__(kThreads) void packed_gen_kernel(GenLayout L, uint32_t n, ModParams mod,
                                                              MontParams mont, MatArg M, DrbgKey key,
                                                              uint64_t chunks, uint64_t batches) {
    constexpr int KT = K + T;
    uint64_t p, chunk;
    split_item(blockIdx.x, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    const bool in0 = b0 < batches, in1 = b0 + 1 < batches;

    uint64_t v0[KT], v1[KT];
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const uint64_t e0 = b0 * K;                       // first secret of this lane
    if (VEC && e0 + 2 * K <= L.len) {                 // interior: K 16-byte loads
        uint64_t tmp[2 * K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ll2 v = load2(sp + e0 + 2 * i);
            tmp[2 * i] = canon_i64(v.x, mod.m, mod.mu);
            tmp[2 * i + 1] = canon_i64(v.y, mod.m, mod.mu);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) { v0[i] = tmp[i]; v1[i] = tmp[K + i]; }
    } else {                                          // ragged tail: zero padding (batched.rs:37-43)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint64_t a = e0 + i, b = e0 + K + i;
            v0[i] = a < L.len ? canon_i64(sp[a], mod.m, mod.mu) : 0;
            v1[i] = b < L.len ? canon_i64(sp[b], mod.m, mod.mu) : 0;
        }
    }

    if (L.rand) {
        const int64_t* rp = L.rand + p * L.rand_stride;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            v0[K + i] = in0 ? canon_i64(rp[b0 * T + i], mod.m, mod.mu) : 0;
            v1[K + i] = in1 ? canon_i64(rp[(b0 + 1) * T + i], mod.m, mod.mu) : 0;
        }
    } else {
        const uint64_t stream = L.first_participant + p;
        const QuadCol qc = quad_col(key);
#pragma unroll
        for (int i = 0; i < T; ++i) drbg_pair<ROUNDS>(key, qc, stream, pair, T, i, mod, v0[K + i], v1[K + i]);
    }

    int64_t* op = L.out + p * L.out_stride_participant + b0;
    const uint32_t direct = L.rand ? 0u : L.direct_rows;                     // 0 or T (systematic share map)
    if (direct) {
#pragma unroll
        for (int i = 0; i < T; ++i) store_pair<VEC>(op + (size_t)i * L.out_stride_clerk, v0[K + i], v1[K + i], in0, in1);
    }
    for (uint32_t j = direct; j < n; ++j) {
        const uint64_t* row = &M.e[(size_t)(j - direct) * KT];
        const uint64_t a = mont_dot<KT>(row, v0, mont.p, mont.pinv);
        const uint64_t b = mont_dot<KT>(row, v1, mont.p, mont.pinv);
        store_pair<VEC>(op + (size_t)j * L.out_stride_clerk, a, b, in0, in1);
    }
}

// -------------------------------------------------------------------------------------------------
// K2'  packed-Shamir share generation, the VALU-lean form (all compiled (k, t) shapes).
//
// gfx950 cost model (tools/microbench_valu.hip): only v_add/sub/xor/and/mov issue at ~2.5 cycles per
// wave64; EVERYTHING else - multiplies, v_mad_*64*, 64-bit adds, shifts, compares, carry ops - costs
// ~4.5.  So the work is organised to minimise instruction count, not multiplier width:
//   * every residue is centred to (-p/2, p/2) and split into balanced signed limbs x = x1*B + x0,
//     B = 2^31, |x0|,|x1| <= 2^30; the matrix constants (Montgomery form, R = B^2 = 2^62) likewise;
//   * a dot product of up to 4 terms is then three signed 64-bit column sums C0, C1, C2 built by 16
//     v_mad_i64_i32 with NO carry handling: |C0|,|C2| <= 2^62, |C1| < 2^63 (longer dot products are
//     cut into groups of 4 terms whose reduced results are added lazily in [0, 2p));
//   * Montgomery reduction runs directly on the columns, one radix-B digit at a time
//     (q = C * (-p^-1) mod B, balanced), again carry-free; the result lies in (-1.5p, 1.5p),
//     +2p and two conditional subtractions make it canonical.
// Bounds and exactness were checked exhaustively in big-int arithmetic (tests/test_limb31_model.py).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t sext31(uint32_t x) { return ((int32_t)(x << 1)) >> 1; }

__device__ __forceinline__ void centre_limbs(uint64_t v, const L31Params& P, int32_t& l0, int32_t& l1) {
    const int64_t c = (int64_t)(v >= P.h ? v - P.p : v);            // |c| <= (p-1)/2 < 2^61
    l0 = sext31((uint32_t)c);
    l1 = (int32_t)(c >> 31) + (int32_t)(((uint32_t)c >> 30) & 1u);     // (c - l0) / B
}

// One-instruction wrappers: hipcc expands an SGPR-resident i32 (sign-extended in the SALU) times a VGPR
// i32 into a 64x32 schoolbook product (v_mul_lo_u32 + v_mad_u64_u32 + ...) instead of the single
// v_mad_i64_i32 it is.  The wrappers pin the instruction; scheduling and allocation stay with hipcc.
__device__ __forceinline__ int64_t mad_sv(int32_t s_a, int32_t v_b, int64_t c) {      // SGPR * VGPR + acc
    int64_t d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "s"(s_a), "v"(v_b), "v"(c) : "vcc");
    return d;
}
__device__ __forceinline__ int64_t mul_sv(int32_t s_a, int32_t v_b) {                 // SGPR * VGPR
    int64_t d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(d) : "s"(s_a), "v"(v_b) : "vcc");
    return d;
}

// One group of N <= 4 terms: carry-free columns + radix-B Montgomery reduction -> value in (-1.5p, 1.5p)
// congruent to  sum_i M_i * v_i  (mod p).
template <int N>
__device__ __forceinline__ int64_t l31_group(const uint64_t* __restrict__ row, const int32_t* v0, const int32_t* v1,
                                             const L31Params& P) {
    static_assert(N >= 1 && N <= 4, "column bounds hold for at most 4 terms");
    int64_t C0, C1, C2;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int32_t m0 = (int32_t)(uint32_t)row[i];
        const int32_t m1 = (int32_t)(uint32_t)(row[i] >> 32);
        if (i == 0) {
            C0 = mul_sv(m0, v0[i]);
            C1 = mul_sv(m0, v1[i]);
            C2 = mul_sv(m1, v1[i]);
        } else {
            C0 = mad_sv(m0, v0[i], C0);
            C1 = mad_sv(m0, v1[i], C1);
            C2 = mad_sv(m1, v1[i], C2);
        }
        C1 = mad_sv(m1, v0[i], C1);
    }
    const int32_t q0 = sext31((uint32_t)C0 * P.pinvB);
    C0 = mad_sv(P.p0, q0, C0);                                  // == 0 mod B
    int64_t E = mad_sv(P.p1, q0, C0 >> 31);
    const int32_t q1 = sext31(((uint32_t)C1 + (uint32_t)E) * P.pinvB);
    E = mad_sv(P.p0, q1, E);                                    // C1 + E == 0 mod B
    // (C1 + E) / B without a 65-bit sum: the low limbs add up to 0 or B
    return mad_sv(P.p1, q1, C2) + (C1 >> 31) + ((E + 0x7FFFFFFF) >> 31);
}

// Five terms in one group: the two cross columns m0*v1 and m1*v0 are kept apart (a shared column would hold ten products
// of up to 2^60: past 2^63), every column stays below 1.25 * 2^62 and the result in (-1.75p, 1.75p) still fits 64 bits -
// six or more terms would not (each term adds up to p/4 to the result).  One reduction per five terms instead of per
// four: (8,2,26) takes 2 groups instead of 3, (8,7,26) 3 instead of 4.  Model: tests/test_limb31_model.py.
__device__ __forceinline__ int64_t l31_group5(const uint64_t* __restrict__ row, const int32_t* v0, const int32_t* v1,
                                              const L31Params& P) {
    int64_t C0, C1a, C1b, C2;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int32_t m0 = (int32_t)(uint32_t)row[i];
        const int32_t m1 = (int32_t)(uint32_t)(row[i] >> 32);
        if (i == 0) {
            C0 = mul_sv(m0, v0[i]); C1a = mul_sv(m0, v1[i]); C1b = mul_sv(m1, v0[i]); C2 = mul_sv(m1, v1[i]);
        } else {
            C0 = mad_sv(m0, v0[i], C0); C1a = mad_sv(m0, v1[i], C1a); C1b = mad_sv(m1, v0[i], C1b); C2 = mad_sv(m1, v1[i], C2);
        }
    }
    const int32_t q0 = sext31((uint32_t)C0 * P.pinvB);
    C0 = mad_sv(P.p0, q0, C0);                                  // == 0 mod B
    int64_t E = mad_sv(P.p1, q0, C0 >> 31);
    const int32_t q1 = sext31(((uint32_t)C1a + (uint32_t)C1b + (uint32_t)E) * P.pinvB);
    E = mad_sv(P.p0, q1, E);                                    // C1a + C1b + E == 0 mod B
    // (C1a + C1b + E) / B without forming the sum (it can pass 2^63): floors of the two columns + the exact quotient of
    // E plus their low limbs
    const uint32_t lows = ((uint32_t)C1a & 0x7FFFFFFFu) + ((uint32_t)C1b & 0x7FFFFFFFu);
    return mad_sv(P.p1, q1, C2) + (C1a >> 31) + (C1b >> 31) + ((E + (int64_t)(uint64_t)lows) >> 31);
}

// ---- the three-digit form (round 4): groups of up to SEVEN terms, ONE reduction per dot product ------------------------------
// A two-digit reduction (R = 2^62) leaves (X + q p) / 2^62 with |X| up to terms x p^2 / 4: more than five terms do not fit a
// signed 64-bit register, so a 7-term dot product paid two reductions, a 10-term one two reductions of five, plus the lazy
// recombination of their results.  With R = 2^93 - a THIRD radix-2^31 digit - the reduced value is X / 2^93 + q p / 2^93, i.e.
// within (-p/2 - eps, p/2 + eps) for ANY number of terms; what limits a group is then only the columns themselves: seven
// products of 2^60 per signed 64-bit column with the two cross columns kept apart.  Between groups the columns are not
// reduced but NORMALISED - carries pushed one column up, the low 31 bits kept, a fifth column C3 of weight 2^93 collecting
// the top - 15 instructions instead of a 24-instruction reduction + select + lazy add.  One dot product of 7 terms: 28
// multiply-adds + 36 (was 28 + 51); of 10 terms: 40 + 49 (was 40 + 57).  The constants of these shapes carry R = 2^93
// (packed_l31_r_bits).  Every register bound, incl. the worst-case prime below 2^62: tests/test_limb31_r93_model.py.
struct L31Cols {
    int64_t C0, C1a, C1b, C2, C3;
};

// N more terms into the columns.  FRESH: the very first group (every column starts with a product); otherwise only C1b
// starts fresh (the normalisation folded it into C1a).
template <int N, bool FRESH>
__device__ __forceinline__ void l31_cols_add(L31Cols& c, const uint64_t* __restrict__ row, const int32_t* v0, const int32_t* v1) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int32_t m0 = (int32_t)(uint32_t)row[i];
        const int32_t m1 = (int32_t)(uint32_t)(row[i] >> 32);
        if (FRESH && i == 0) {
            c.C0 = mul_sv(m0, v0[i]); c.C1a = mul_sv(m0, v1[i]); c.C2 = mul_sv(m1, v1[i]);
        } else {
            c.C0 = mad_sv(m0, v0[i], c.C0); c.C1a = mad_sv(m0, v1[i], c.C1a); c.C2 = mad_sv(m1, v1[i], c.C2);
        }
        c.C1b = i == 0 ? mul_sv(m1, v0[i]) : mad_sv(m1, v0[i], c.C1b);
    }
}

template <bool FIRST>
__device__ __forceinline__ void l31_normalize(L31Cols& c) {
    const int64_t t0 = c.C0 >> 31;
    c.C0 = (int64_t)(uint64_t)((uint32_t)c.C0 & 0x7FFFFFFFu);
    c.C1a += t0;
    const int64_t hi = (c.C1a >> 31) + (c.C1b >> 31);
    const uint32_t lows = ((uint32_t)c.C1a & 0x7FFFFFFFu) + ((uint32_t)c.C1b & 0x7FFFFFFFu);      // < 2^32
    c.C1a = (int64_t)(uint64_t)lows;                                                            // C1b restarts with the next group
    c.C2 += hi;
    const int64_t t2 = c.C2 >> 31;
    c.C2 = (int64_t)(uint64_t)((uint32_t)c.C2 & 0x7FFFFFFFu);
    c.C3 = FIRST ? t2 : c.C3 + t2;
}

// columns -> sum * 2^-93 mod p, canonical.  SPLIT0: the last group held seven terms, C0 + q0 p0 can pass 2^63 and its quotient
// by 2^31 is formed from the floor of C0 and the exact quotient of (low limb + q0 p0).  HAS_C3: more than one group.
template <bool SPLIT0, bool HAS_C3>
__device__ __forceinline__ uint64_t l31_redc3(const L31Cols& c, const L31Params& P) {
    const int32_t q0 = sext31((uint32_t)c.C0 * P.pinvB);
    int64_t d0;
    if (SPLIT0) d0 = (c.C0 >> 31) + (mad_sv(P.p0, q0, (int64_t)(uint64_t)((uint32_t)c.C0 & 0x7FFFFFFFu)) >> 31);
    else d0 = mad_sv(P.p0, q0, c.C0) >> 31;
    const int64_t E1 = mad_sv(P.p1, q0, d0);
    const uint32_t lows = ((uint32_t)c.C1a & 0x7FFFFFFFu) + ((uint32_t)c.C1b & 0x7FFFFFFFu);
    const int32_t q1 = sext31(((uint32_t)c.C1a + (uint32_t)c.C1b + (uint32_t)E1) * P.pinvB);
    const int64_t F = mad_sv(P.p0, q1, E1);                                  // C1a + C1b + F == 0 mod 2^31
    const int64_t carry1 = (c.C1a >> 31) + (c.C1b >> 31) + ((F + (int64_t)(uint64_t)lows) >> 31);
    const int64_t G = mad_sv(P.p1, q1, (int64_t)(uint64_t)((uint32_t)c.C2 & 0x7FFFFFFFu)) + carry1;
    const int32_t q2 = sext31((uint32_t)G * P.pinvB);
    const int64_t H = mad_sv(P.p0, q2, G);                                   // == 0 mod 2^31
    int64_t top = (c.C2 >> 31) + (H >> 31);
    if (HAS_C3) top += c.C3;
    const int64_t res = mad_sv(P.p1, q2, top);                               // in (-p, p)
    const uint64_t lifted = (uint64_t)res + P.p;                             // wraps exactly when res < 0
    return lifted < (uint64_t)res ? lifted : (uint64_t)res;
}

// WIDE (round 5): the whole dot product of 9 .. 12 terms as ONE group - no normalisation at all (10 terms: 40 multiply-adds + 34
// instead of 40 + 49).  Ten products of 2^60 pass a signed 64-bit column only if the constant limbs are large all at once;
// for random constants sum |limb| is ~ KT / 2 x 2^30 against the 8 x 2^30 a column can take, so the host admits the form on
// the ACTUAL constants of both share maps (l31_wide_group_ok in sda_capi.cpp -> L31Params::wide) and the 7 + rest form
// serves the handle otherwise.  BASELINE config 4's (8,2,26) over the 62-bit prime: worst row 6.93 x 2^30.
template <int KT, bool WIDE = false>
__device__ __forceinline__ uint64_t l31_dot3(const uint64_t* __restrict__ row, const int32_t (&v0)[KT], const int32_t (&v1)[KT],
                                             const L31Params& P) {
    if constexpr (WIDE) {
        static_assert(KT >= 9 && KT <= 12, "the wide group is compiled for 9 .. 12 terms");
        L31Cols c;
        l31_cols_add<KT, true>(c, row, v0, v1);
        return l31_redc3<true, false>(c, P);
    }
    // groups of seven; a remainder of ONE term joins the group before it (15 = 7 + 8: saves a normalisation) - eight products
    // of 2^60 can pass 2^63 only if all eight limbs are -2^30, which the host rules out on the actual constants
    // (l31_eight_term_group_ok in sda_capi.cpp; the shape is served by another kernel otherwise)
    constexpr bool EIGHT = KT > 7 && KT % 7 == 1;
    constexpr int FULL = EIGHT ? KT / 7 - 1 : KT / 7, REST = EIGHT ? 8 : KT % 7;
    static_assert(FULL >= 1 || REST >= 1, "empty dot product");
    L31Cols c;
    if constexpr (FULL == 0) {
        l31_cols_add<REST, true>(c, row, v0, v1);
        return l31_redc3<(REST >= 7), false>(c, P);
    } else {
        l31_cols_add<7, true>(c, row, v0, v1);
#pragma unroll
        for (int g = 1; g < FULL; ++g) {
            if (g == 1) l31_normalize<true>(c); else l31_normalize<false>(c);
            l31_cols_add<7, false>(c, row + 7 * g, v0 + 7 * g, v1 + 7 * g);
        }
        if constexpr (REST > 0) {
            if (FULL == 1) l31_normalize<true>(c); else l31_normalize<false>(c);
            l31_cols_add<REST, false>(c, row + 7 * FULL, v0 + 7 * FULL, v1 + 7 * FULL);
            return l31_redc3<(REST >= 7), true>(c, P);
        } else {
            return FULL > 1 ? l31_redc3<true, true>(c, P) : l31_redc3<true, false>(c, P);
        }
    }
}

// The wide group in KARATSUBA form (round 6): three multiply-adds per term instead of four.  With ms = m0 + m1 and vs = v0 + v1 (both fit a
// signed 32-bit register: a low limb lies in [-2^30, 2^30 - 1], a high limb in [-2^30, 2^30]) the cross products of a term are
// ms vs - m0 v0 - m1 v1, so a group needs the columns C0, C2 and ONE middle column M = sum ms vs, and its cross column is
// M - C0 - C2 - exact in 64-bit wrap-around arithmetic whenever the cross column itself fits, whatever M overflowed to.  The
// cross column of the WHOLE dot product does not fit (that is why the plain form keeps C1a and C1b apart); the cross column of half
// of it does for the constants the host admits (l31_karatsuba_ok: sum over the half of |m0| + |m1|, times the 2^30 a value's limb
// can reach, stays below 2^63 - BASELINE config 4's worst half 7.07 x 2^30 of the 8 x 2^30 allowed; any constants: three terms).
// So the terms are split in two halves with their own C0, C2, M;
// the halves' cross columns take the places of C1a and C1b (the reduction only ever uses their sum, in split form), C0 and C2 are
// the sums of the halves' (within 2^63 by the host's check on the actual constants, l31_wide_group_ok, as before).  Ten terms: 30
// multiply-adds + 34 + 12 additions instead of 40 + 34; ms is a scalar addition (the constants live in SGPRs), vs is formed once
// per batch for all rows.  Model: tests/test_limb31_r93_model.py (test_karatsuba_halves_*).
template <int KT>
__device__ __forceinline__ uint64_t l31_dot3_wide_k(const uint64_t* __restrict__ row, const int32_t (&v0)[KT], const int32_t (&v1)[KT],
                                                    const int32_t (&vs)[KT], const L31Params& P) {
    static_assert(KT >= 9 && KT <= 12, "the wide group is compiled for 9 .. 12 terms");
    constexpr int HALF = (KT + 1) / 2;
    int64_t C0[2], C2[2], M[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = h * HALF; i < (h ? KT : HALF); ++i) {
            const int32_t m0 = (int32_t)(uint32_t)row[i];
            const int32_t m1 = (int32_t)(uint32_t)(row[i] >> 32);
            const int32_t ms = m0 + m1;
            if (i == h * HALF) { C0[h] = mul_sv(m0, v0[i]); C2[h] = mul_sv(m1, v1[i]); M[h] = mul_sv(ms, vs[i]); }
            else { C0[h] = mad_sv(m0, v0[i], C0[h]); C2[h] = mad_sv(m1, v1[i], C2[h]); M[h] = mad_sv(ms, vs[i], M[h]); }
        }
    }
    L31Cols c;
    c.C1a = (int64_t)((uint64_t)M[0] - (uint64_t)C0[0] - (uint64_t)C2[0]);
    c.C1b = (int64_t)((uint64_t)M[1] - (uint64_t)C0[1] - (uint64_t)C2[1]);
    c.C0 = C0[0] + C0[1];
    c.C2 = C2[0] + C2[1];
    c.C3 = 0;
    return l31_redc3<true, false>(c, P);
}

// which compiled (k, t) shapes use the three-digit form: those with a compiled instance in BOTH the plain and the dual-role
// launchers (every other shape may be served by the run-time (k, t) kernels, whose constants carry R = 2^62) and a term
// count it pays for
template <int K, int T>
struct L31UseR93 {
    static constexpr bool value = (K == 3 && T == 4) || (K == 8 && T == 2) || (K == 8 && T == 7);
};

// x in [0, 2m) -> [0, m) given neg_m = 2^64 - m: y = x - m wraps (y > x) exactly when x < m.  One 64-bit add, one compare
// and the select - four VALU instructions; the `x >= m ? x - m : x` form costs hipcc seven (it materialises m in VGPRs
// for a masked subtraction).
__device__ __forceinline__ uint64_t condsub_neg(uint64_t x, uint64_t neg_m) {
    const uint64_t y = x + neg_m;
    return y < x ? y : x;
}

// sum_i M_i * v_i mod p for any KT: groups of <= 4 terms - or of 5 where that saves a group (KT = 5, 9, 10, 13, 14, 15,
// ...) - partial results kept lazily in [0, 2p)
template <int KT>
__device__ __forceinline__ uint64_t l31_dot(const uint64_t* __restrict__ row, const int32_t (&v0)[KT],
                                            const int32_t (&v1)[KT], const L31Params& P) {
    constexpr int G4 = (KT + 3) / 4, G5 = (KT + 4) / 5;
    constexpr int FIVES = G5 < G4 ? KT - 4 * G5 : 0;           // that many groups of five, the rest of four (or fewer)
    uint64_t r = 0;
    int g = 0;
#pragma unroll
    for (int grp = 0; grp < (G5 < G4 ? G5 : G4); ++grp) {
        int64_t top;
        const int left = KT - g;
        int used;
        if (grp < FIVES) { top = l31_group5(row + g, v0 + g, v1 + g, P); used = 5; }
        else if (left >= 4) { top = l31_group<4>(row + g, v0 + g, v1 + g, P); used = 4; }
        else if (left == 3) { top = l31_group<3>(row + g, v0 + g, v1 + g, P); used = 3; }
        else if (left == 2) { top = l31_group<2>(row + g, v0 + g, v1 + g, P); used = 2; }
        else { top = l31_group<1>(row + g, v0 + g, v1 + g, P); used = 1; }
        const uint64_t lifted = (uint64_t)top + P.p2;                     // top in (-1.75p, 1.75p): wraps exactly when top < 0
        const uint64_t u = lifted < (uint64_t)top ? lifted : (uint64_t)top;  // [0, 2p)
        r = grp == 0 ? u : condsub_neg(r + u, P.np2);                     // [0, 2p)
        g += used;
    }
    return condsub_neg(r, P.np);
}

template <int K, int T, int ROUNDS, bool VEC>
__device__ __forceinline__ void packed_gen_l31_body(const GenLayout& L, uint32_t n, const ModParams& mod,
                                                    const L31Params& lp, const MatArg& M, const DrbgKey& key,
                                                    uint64_t chunks, uint64_t batches, uint64_t item) {
    constexpr int KT = K + T;
    uint64_t p, chunk;
    split_item(item, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    const bool in0 = b0 < batches, in1 = b0 + 1 < batches;

    uint64_t s0[KT], s1[KT];
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const uint64_t e0 = b0 * K;
    if (VEC && e0 + 2 * K <= L.len) {
        uint64_t tmp[2 * K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ll2 v = load2(sp + e0 + 2 * i);
            tmp[2 * i] = canon_i64(v.x, mod.m, mod.mu);
            tmp[2 * i + 1] = canon_i64(v.y, mod.m, mod.mu);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) { s0[i] = tmp[i]; s1[i] = tmp[K + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint64_t a = e0 + i, b = e0 + K + i;
            s0[i] = a < L.len ? canon_i64(sp[a], mod.m, mod.mu) : 0;
            s1[i] = b < L.len ? canon_i64(sp[b], mod.m, mod.mu) : 0;
        }
    }
    if (L.rand) {
        const int64_t* rp = L.rand + p * L.rand_stride;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            s0[K + i] = in0 ? canon_i64(rp[b0 * T + i], mod.m, mod.mu) : 0;
            s1[K + i] = in1 ? canon_i64(rp[(b0 + 1) * T + i], mod.m, mod.mu) : 0;
        }
    } else {
        const uint64_t stream = L.first_participant + p;
        const QuadCol qc = quad_col(key);
#pragma unroll
        for (int i = 0; i < T; ++i) drbg_pair<ROUNDS>(key, qc, stream, pair, T, i, mod, s0[K + i], s1[K + i]);
    }

    int32_t a0[KT], a1[KT], c0[KT], c1[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        centre_limbs(s0[i], lp, a0[i], a1[i]);
        centre_limbs(s1[i], lp, c0[i], c1[i]);
    }

    int64_t* op = L.out + p * L.out_stride_participant + b0;
    const uint32_t direct = L.rand ? 0u : L.direct_rows;                     // 0 or T (systematic share map)
    if (direct) {
#pragma unroll
        for (int i = 0; i < T; ++i) store_pair<VEC>(op + (size_t)i * L.out_stride_clerk, s0[K + i], s1[K + i], in0, in1);
    }
    if constexpr (L31UseR93<K, T>::value && KT >= 9 && KT <= 12) {
        if (lp.wide == 2) {                                                  // uniform: ... and its Karatsuba form
            int32_t as[KT], cs[KT];                                          // v0 + v1 of every term: the third operand
#pragma unroll
            for (int i = 0; i < KT; ++i) { as[i] = a0[i] + a1[i]; cs[i] = c0[i] + c1[i]; }
            for (uint32_t j = direct; j < n; ++j) {
                const uint64_t* row = &M.e[(size_t)(j - direct) * KT];
                const uint64_t a = l31_dot3_wide_k<KT>(row, a0, a1, as, lp);
                const uint64_t b = l31_dot3_wide_k<KT>(row, c0, c1, cs, lp);
                store_pair<VEC>(op + (size_t)j * L.out_stride_clerk, a, b, in0, in1);
            }
            return;
        }
        if (lp.wide) {                                                       // uniform: the host admitted the one-group form
            for (uint32_t j = direct; j < n; ++j) {
                const uint64_t* row = &M.e[(size_t)(j - direct) * KT];
                const uint64_t a = l31_dot3<KT, true>(row, a0, a1, lp);
                const uint64_t b = l31_dot3<KT, true>(row, c0, c1, lp);
                store_pair<VEC>(op + (size_t)j * L.out_stride_clerk, a, b, in0, in1);
            }
            return;
        }
    }
    for (uint32_t j = direct; j < n; ++j) {
        const uint64_t* row = &M.e[(size_t)(j - direct) * KT];
        uint64_t a, b;
        if constexpr (L31UseR93<K, T>::value) {
            a = l31_dot3<KT>(row, a0, a1, lp);
            b = l31_dot3<KT>(row, c0, c1, lp);
        } else {
            a = l31_dot<KT>(row, a0, a1, lp);
            b = l31_dot<KT>(row, c0, c1, lp);
        }
        store_pair<VEC>(op + (size_t)j * L.out_stride_clerk, a, b, in0, in1);
    }
}

// Register caps: one more wave per SIMD than hipcc's free allocation gives (k + t >= 12: 128 VGPRs = 4 waves instead of
// 137 = 3; k + t >= 6: 96 = 5 waves instead of 101-113 = 4).  Measured, interleaved A/B on one box (tools/r02_ab_lb.sh):
// (3,4,8) +3.4 %, (8,7,26) +3.6 %, (8,2,26) +2.5 % on the dual-role launch - the kernels are VALU-bound and the extra
// wave hides more of the s_load / dependent-issue latency; the few spilled dwords it costs (32-64 B per lane) do not show.
#ifndef SDA_LB_FOUR_FROM
#define SDA_LB_FOUR_FROM 12
#endif
#define SDA_LB(K_, T_) __launch_bounds__(kThreads, ((K_) + (T_) >= SDA_LB_FOUR_FROM ? 4 : ((K_) + (T_) >= 6 ? 5 : 1)))
template <int K, int T, int ROUNDS, bool VEC>
__global__ SDA_LB(K, T) void packed_gen_l31_kernel(GenLayout L, uint32_t n, ModParams mod,
                                                                  L31Params lp, MatArg M, DrbgKey key,
                                                                  uint64_t chunks, uint64_t batches) {
    packed_gen_l31_body<K, T, ROUNDS, VEC>(L, n, mod, lp, M, key, chunks, batches, blockIdx.x);
}

// -------------------------------------------------------------------------------------------------
// K2''  packed-Shamir share generation as a LIMB GEMM on the matrix cores.
//
// For the shapes whose n x (k + t) mat-vec is VALU-bound (k + t ~ 10..16, n = 26: 60..80 VALU wave-instructions per
// batch in the limb-31 kernel, four v_mad_i64_i32 per term) the products are moved to v_mfma_i32_16x16x64_i8:
//   * every residue x in [0, p) is written in BALANCED base-256 digits, x = sum d_i 256^i with d_i in [-128, 127]:
//     the bytes of (x + 0x8080..80) xor 0x8080..80 - two 64-bit instructions, and the eight digits of a value ARE the
//     eight bytes of a register pair; the matrix constants (Montgomery form, R = 2^64) likewise, on the host;
//   * column c = sum over terms and i + l = c of dM_i dV_l is one row of a 16 x 64 times 64 x 16 integer product: the B
//     operand of lane (batch = lane & 15, g = lane >> 4) is simply the 16 bytes of the values of terms 2g, 2g + 1 of that
//     batch; the A operand of lane (c = lane & 15, g) is a Toeplitz row - byte l' = constant byte c - l' - cut out of the
//     8-byte constant with two v_perm_b32 under per-lane selectors (no 1 KiB fragment table per clerk);  |column| < 2^21;
//   * the 15 columns of an output land four per lane on the four lanes (batch, g = 0..3).  A wave works on 64 batches =
//     four 16-batch tiles, so after three v_mad_i64_i32 per tile a 4 x 4 transpose across the wave's four 16-lane rows
//     (v_permlane32_swap / v_permlane16_swap, eight instructions) hands every lane the four partial sums of ONE batch:
//     a 128-bit assemble, two conditional subtractions and one Montgomery reduction later it stores one share per
//     clerk, lane-contiguous.
// Values reach the operand layout through a wave-private LDS tile [batch][term] (rows padded to 144 B: both the 8-byte
// writes and the 16-byte operand reads spread over all banks).  The draws are the same sda-drbg-v1 quad blocks.
// -------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
static constexpr uint64_t kBalancedBias = 0x8080808080808080ull;
__device__ __forceinline__ uint64_t balanced_bytes(uint64_t x) { return (x + kBalancedBias) ^ kBalancedBias; }   // x: two's complement
// canonical residue -> its representative in (-p/2, p/2] (two's complement): halves the magnitude of every product
__device__ __forceinline__ uint64_t centred(uint64_t v, uint64_t p) { return v > (p >> 1) ? v - p : v; }
__device__ __forceinline__ void rows_swap32(uint32_t& a, uint32_t& b) {     // a rows 2,3 <-> b rows 0,1 (rows of 16 lanes)
    const v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r.x; b = r.y;
}
__device__ __forceinline__ void rows_swap16(uint32_t& a, uint32_t& b) {     // a rows 1,3 <-> b rows 0,2
    const v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r.x; b = r.y;
}

__device__ __forceinline__ int64_t vmad_i64(int32_t a, int32_t b, int64_t c) {      // a * b + c: one v_mad_i64_i32
    int64_t d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}

static constexpr int kMfmaMaxClerks = 242;                   // LDS table of constants: n x 8 KS x 8 B, dynamic (242 clerks x 128 B = 31 KiB beside the 37 KiB of tiles: two workgroups per CU)
static constexpr int kMfmaWaveBatches = 64;

// COUNT consecutive values per batch from src[first + e], e < 64 * COUNT (lane-contiguous 8-byte loads) -> tile[batch][term0 + ..]
template <int COUNT, int ROWDW>
__device__ __forceinline__ void mfma_stage_values(uint32_t* tile, const int64_t* __restrict__ src, uint64_t first, uint64_t src_len,
                                                  int term0, const ModParams& mod) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int q = 0; q < COUNT; ++q) {
        const uint32_t e = lane + 64u * q;
        const uint64_t idx = first + e;
        const uint64_t v = idx < src_len ? canon_i64(src[idx], mod.m, mod.mu) : 0;       // zero padding (batched.rs:37-43)
        const uint32_t bl = e / COUNT, tm = e - bl * COUNT;
        *reinterpret_cast<uint64_t*>(tile + bl * ROWDW + 2 * (term0 + tm)) = balanced_bytes(centred(v, mod.m));
    }
}

// the same with a run-time count (shapes without a compiled instance)
template <int ROWDW>
__device__ __forceinline__ void mfma_stage_values_rt(uint32_t* tile, const int64_t* __restrict__ src, uint64_t first, uint64_t src_len,
                                                     uint32_t count, uint32_t term0, const ModParams& mod) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t q = 0; q < count; ++q) {
        const uint32_t e = lane + 64u * q;
        const uint64_t idx = first + e;
        const uint64_t v = idx < src_len ? canon_i64(src[idx], mod.m, mod.mu) : 0;
        const uint32_t bl = e / count, tm = e - bl * count;
        *reinterpret_cast<uint64_t*>(tile + bl * ROWDW + 2 * (term0 + tm)) = balanced_bytes(centred(v, mod.m));
    }
}

// the 16 x 64 x 16 products of clerk j with the four batch tiles: d[tile] = 4 of the 15 columns (rows 4 g .. 4 g + 3).
// A operand of lane (c = lane & 15, g): byte l' of term 2 g (+ 1) = byte c - l' of that constant, 0 outside 0..7 - a Toeplitz
// row, cut out of the 8-byte constant with two v_perm_b32 under per-lane selectors.
template <int KS>
__device__ __forceinline__ void mfma_clerk_products(v4i (&d)[4], const uint64_t* mconst, uint32_t j, uint32_t g, uint32_t sel_lo,
                                                    uint32_t sel_hi, const v4i (&bfrag)[4][KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const ull2 m2 = *reinterpret_cast<const ull2*>(&mconst[(j * KS + ks) * 8 + 2 * g]);
        v4i a;
        a.x = (int)__builtin_amdgcn_perm((uint32_t)(m2.x >> 32), (uint32_t)m2.x, sel_lo);
        a.y = (int)__builtin_amdgcn_perm((uint32_t)(m2.x >> 32), (uint32_t)m2.x, sel_hi);
        a.z = (int)__builtin_amdgcn_perm((uint32_t)(m2.y >> 32), (uint32_t)m2.y, sel_lo);
        a.w = (int)__builtin_amdgcn_perm((uint32_t)(m2.y >> 32), (uint32_t)m2.y, sel_hi);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const v4i zero = {0, 0, 0, 0};
            d[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bfrag[nt][ks], ks == 0 ? zero : d[nt], 0, 0, 0);
        }
    }
}

// the four tiles' columns of one clerk -> one share per lane (batch = lane of the wave's 64), stored at `o`
__device__ __forceinline__ void mfma_clerk_finish(const v4i (&d)[4], int32_t mul3, const MontParams& mont, bool live, int64_t* o) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int64_t pa = vmad_i64(d[nt].y, 1 << 8, (int64_t)d[nt].x);
        pa = vmad_i64(d[nt].z, 1 << 16, pa);
        pa = vmad_i64(d[nt].w, mul3, pa);
        lo[nt] = (uint32_t)pa; hi[nt] = (uint32_t)((uint64_t)pa >> 32);
    }
    // 4 x 4 transpose (register index = tile, lane row = column group): lane row g ends up with tile g's four partial sums
    rows_swap32(lo[0], lo[2]); rows_swap32(lo[1], lo[3]); rows_swap16(lo[0], lo[1]); rows_swap16(lo[2], lo[3]);
    rows_swap32(hi[0], hi[2]); rows_swap32(hi[1], hi[3]); rows_swap16(hi[0], hi[1]); rows_swap16(hi[2], hi[3]);
    // X = q0 + q1 2^32 + q2 2^64 + q3 2^96 = sum over <= 16 terms of constant * value, both centred: exact, |X| <= 4 p^2 < p 2^64
    const uint64_t q0 = ((uint64_t)hi[0] << 32) | lo[0];
    const uint64_t xlo = q0 + ((uint64_t)lo[1] << 32);
    const int64_t xhi = (int64_t)((int32_t)hi[0] >> 31) + (int64_t)(int32_t)hi[1] + (xlo < q0 ? 1 : 0) +
                        (int64_t)(((uint64_t)hi[2] << 32) | lo[2]) + (int64_t)((uint64_t)lo[3] << 32);
    // signed REDC (R = 2^64): X + (X.lo * -p^-1 mod 2^64) p is a multiple of 2^64; the quotient lies in [-p, 2p)
    const uint64_t m = xlo * mont.pinv;
    int64_t t = xhi + (int64_t)mulhi64(m, mont.p) + (xlo != 0 ? 1 : 0);
    t += (t >> 63) & (int64_t)mont.p;
    const uint64_t share = (uint64_t)t >= mont.p ? (uint64_t)t - mont.p : (uint64_t)t;
    if (live) __builtin_nontemporal_store((long long)share, reinterpret_cast<long long*>(o));
}

template <int K, int T, int ROUNDS>
__device__ __forceinline__ void packed_gen_mfma_body(const GenLayout& L, uint32_t n, const ModParams& mod, const MontParams& mont,
                                                     const uint64_t* __restrict__ Mbal, const DrbgKey& key, uint64_t chunks,
                                                     uint64_t batches, uint32_t iters, uint64_t item, uint32_t k_rt, uint32_t t_rt) {
    // K = 0: k and t are run-time arguments (k + t <= 16, two 64-slot MFMA steps)
    constexpr int KS = K ? (K + T + 7) / 8 : 2, ROWDW = KS * 16 + 4;
    static_assert(K + T <= 16, "two 64-slot MFMA steps hold 16 terms");
    const uint32_t k = K ? (uint32_t)K : k_rt, t = K ? (uint32_t)T : t_rt;
    extern __shared__ __attribute__((aligned(16))) uint64_t mconst[];       // n * KS * 8 constants (dynamic: sized by the clerk count)
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kThreads / 64][kMfmaWaveBatches * ROWDW];
    uint64_t p, chunk;
    split_item(item, chunks, p, chunk);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, col = lane & 15u, g = lane >> 4;
    uint32_t* tile = tiles[wave];
    const uint32_t direct = L.rand ? 0u : L.direct_rows;                     // 0 or t: rows 0..t-1 are the draws (systematic share map)
    const uint32_t n_mat = n - direct;                                       // rows of the constant table
    for (uint32_t i = threadIdx.x; i < n_mat * KS * 8; i += kThreads) mconst[i] = Mbal[i];
    for (uint32_t i = lane; i < (uint32_t)(kMfmaWaveBatches * ROWDW); i += 64) tile[i] = 0;       // unused terms stay zero
    __syncthreads();

    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    int64_t* op = L.out + p * L.out_stride_participant;
    const uint64_t stream = L.first_participant + p;
    const QuadCol qc = quad_col(key);
    // v_perm_b32 selectors of this lane's Toeplitz row c = col: output byte l' takes constant byte c - l' (0x0c = zero)
    uint32_t sel_lo = 0, sel_hi = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int src = (int)col - l;
        const uint32_t sb = (src >= 0 && src <= 7) ? (uint32_t)src : 0x0cu;
        if (l < 4) sel_lo |= sb << (8 * l); else sel_hi |= sb << (8 * (l - 4));
    }
    const int32_t mul3 = g == 3 ? 0 : (1 << 24);                                                 // column 15 does not exist
    // 16-byte stores of the direct rows need even strides and an aligned base (b0 and the pair offset are even)
    const bool pair_vec = ((reinterpret_cast<uintptr_t>(op) & 15u) == 0) && (L.out_stride_clerk % 2 == 0);

    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t b0 = ((chunk * iters + it) * (kThreads / 64) + wave) * kMfmaWaveBatches;   // wave-uniform
        if (b0 >= batches) break;
        // ---- stage [secrets ; draws] of 64 batches in the operand layout -----------------------------------------
        if constexpr (K != 0) mfma_stage_values<K, ROWDW>(tile, sp, b0 * K, L.len, 0, mod);
        else mfma_stage_values_rt<ROWDW>(tile, sp, b0 * k, L.len, k, 0, mod);
        if (rp) {
            if constexpr (K != 0 && T != 0) mfma_stage_values<T, ROWDW>(tile, rp, b0 * T, batches * T, K, mod);
            else mfma_stage_values_rt<ROWDW>(tile, rp, b0 * t, batches * t, t, k, mod);
        } else {
            auto draw_pass = [&](uint32_t pass) {
                const uint32_t blk = 16u * pass + (lane >> 2);          // block of (draw i, group G of 8 batches)
                const uint32_t G = blk & 7u, i = blk >> 3;
                uint64_t r0, r1;
                drbg_pair<ROUNDS>(key, qc, stream, (b0 >> 1) + 4 * G + (lane & 3u), t, i < t ? i : 0, mod, r0, r1);
                if (i < t) {
                    uint32_t* row = tile + (8 * G + 2 * (lane & 3u)) * ROWDW + 2 * (k + i);
                    *reinterpret_cast<uint64_t*>(row) = balanced_bytes(centred(r0, mod.m));
                    *reinterpret_cast<uint64_t*>(row + ROWDW) = balanced_bytes(centred(r1, mod.m));
                    if (direct) {                                            // draw i of these two batches IS their share i
                        const uint64_t bb = b0 + 8 * G + 2 * (lane & 3u);
                        int64_t* o = op + (size_t)i * L.out_stride_clerk + bb;
                        if (pair_vec && bb + 1 < batches) store2(o, r0, r1);
                        else {
                            if (bb < batches) o[0] = (int64_t)r0;
                            if (bb + 1 < batches) o[1] = (int64_t)r1;
                        }
                    }
                }
            };
            if constexpr (K != 0) {
#pragma unroll
                for (int pass = 0; pass < (8 * T + 15) / 16; ++pass) draw_pass((uint32_t)pass);
            } else {
                for (uint32_t pass = 0; pass < (8 * t + 15) / 16; ++pass) draw_pass(pass);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v4i bfrag[4][KS];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                bfrag[nt][ks] = *reinterpret_cast<const v4i*>(tile + (16 * nt + col) * ROWDW + 16 * ks + 4 * g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next iteration's staging stays behind these reads
        __builtin_amdgcn_wave_barrier();

        const bool live = b0 + lane < batches;
        int64_t* orow = op + b0 + lane;
        // two accumulator sets in turn: the next clerk's products are issued before this clerk's are consumed, so the matrix
        // cores work under the VALU tail
        v4i dA[4], dB[4];
        if (n_mat) mfma_clerk_products<KS>(dA, mconst, 0, g, sel_lo, sel_hi, bfrag);
        for (uint32_t j = 0; j < n_mat; j += 2) {                            // table row j = output row direct + j
            if (j + 1 < n_mat) mfma_clerk_products<KS>(dB, mconst, j + 1, g, sel_lo, sel_hi, bfrag);
            mfma_clerk_finish(dA, mul3, mont, live, orow + (size_t)(direct + j) * L.out_stride_clerk);
            if (j + 1 < n_mat) {
                if (j + 2 < n_mat) mfma_clerk_products<KS>(dA, mconst, j + 2, g, sel_lo, sel_hi, bfrag);
                mfma_clerk_finish(dB, mul3, mont, live, orow + (size_t)(direct + j + 1) * L.out_stride_clerk);
            }
        }
    }
}

template <int K, int T, int ROUNDS>
__global__ __launch_bounds__(kThreads, 3) void packed_gen_mfma_kernel(GenLayout L, uint32_t n, ModParams mod, MontParams mont,
                                                                      const uint64_t* __restrict__ Mbal, DrbgKey key,
                                                                      uint64_t chunks, uint64_t batches, uint32_t iters,
                                                                      uint32_t k_rt, uint32_t t_rt) {
    packed_gen_mfma_body<K, T, ROUNDS>(L, n, mod, mont, Mbal, key, chunks, batches, iters, blockIdx.x, k_rt, t_rt);
}

// ---- run-time (k, t): the same arithmetic for shapes that have no compiled instance ---------------------------------
// k and t are kernel arguments, KTMAX (4 / 8 / 12 / 16) bounds k + t.  The value limbs of the terms beyond k + t are
// zero and every group of four terms is either done in full or skipped by a wave-uniform branch, so a dot product
// costs what the next multiple of four terms costs; the matrix row may be read up to three entries past its end
// (MatArg is zero-padded and n (k + t) + 6 <= SDA_MAT_ARG_MAX is required: the three-digit form reads up to six).
// Round 5: from 17 terms on (the global-matrix kernels for k + t <= 32 / <= 64) the run-time kernels use the THREE-digit form too
// (R = 2^93: groups of seven terms, a carry normalisation between groups, ONE reduction per dot product) - a 33-term dot product
// is 5 x 28 multiply-adds + 4 x 15 + 36 instructions instead of 9 x (16 + 21): (20,13,80) 11.9 -> 15.2, (20,11,40) 50.9 -> 62,
// (10,7,26) 52.8 -> 61.3 Gelem/s (interleaved A/B of two builds, profiles/r05/ab_runtime_three_digit*.txt).  The groups a dot product
// does not reach are skipped by wave-uniform branches; the last group may read up to six row entries past the end (zero-padded
// constants, zero value limbs).  Up to 16 terms the two-digit form stays: measured, (9,6,26) ran 18 % SLOWER in the three-digit
// form (79 -> 65 Gelem/s: at 184 registers the longer live ranges of five columns cost more than two reductions save).
// packed_l31_r_bits() tells the host which radix a shape's constants carry.  Model: tests/test_limb31_r93_model.py.
template <int KTMAX, int G>
__device__ __forceinline__ void l31_rt3_groups(L31Cols& c, const uint64_t* __restrict__ row, const int32_t* v0, const int32_t* v1, uint32_t kt) {
    if constexpr (G < KTMAX) {
        if ((uint32_t)G < kt) {                                               // wave-uniform
            l31_normalize<false>(c);
            l31_cols_add<(KTMAX - G < 7 ? KTMAX - G : 7), false>(c, row + G, v0 + G, v1 + G);
        }
        l31_rt3_groups<KTMAX, G + 7>(c, row, v0, v1, kt);
    }
}
template <int KTMAX>
__device__ __forceinline__ uint64_t l31_dot_rt3(const uint64_t* __restrict__ row, const int32_t (&v0)[KTMAX],
                                                const int32_t (&v1)[KTMAX], uint32_t kt, const L31Params& P) {
    L31Cols c;
    c.C3 = 0;
    l31_cols_add<(KTMAX < 7 ? KTMAX : 7), true>(c, row, v0, v1);
    l31_rt3_groups<KTMAX, 7>(c, row, v0, v1, kt);
    return l31_redc3<true, true>(c, P);
}

template <int KTMAX>
__device__ __forceinline__ uint64_t l31_dot_rt(const uint64_t* __restrict__ row, const int32_t (&v0)[KTMAX],
                                               const int32_t (&v1)[KTMAX], uint32_t kt, const L31Params& P) {
    if constexpr (KTMAX >= 32) return l31_dot_rt3<KTMAX>(row, v0, v1, kt, P);
    uint64_t r = 0;
#pragma unroll
    for (int g = 0; g < KTMAX; g += 4) {
        if ((uint32_t)g < kt) {
            const int64_t top = l31_group<4>(row + g, v0 + g, v1 + g, P);
            const uint64_t lifted = (uint64_t)top + P.p2;
            const uint64_t u = lifted < (uint64_t)top ? lifted : (uint64_t)top;
            r = g == 0 ? u : condsub_neg(r + u, P.np2);
        }
    }
    return condsub_neg(r, P.np);
}

template <int KTMAX, int ROUNDS>
__device__ __forceinline__ void packed_gen_l31_rt_body(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t,
                                                       const ModParams& mod, const L31Params& lp,
                                                       const uint64_t* __restrict__ Mrows, const DrbgKey& key,
                                                       uint64_t chunks, uint64_t batches, bool vec, uint64_t item) {
    const uint32_t kt = k + t;
    uint64_t p, chunk;
    split_item(item, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    const bool in0 = b0 < batches, in1 = b0 + 1 < batches;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    const uint64_t e0 = b0 * k;
    const uint64_t stream = L.first_participant + p;
    const QuadCol qc = quad_col(key);
    int64_t* op = L.out + p * L.out_stride_participant + b0;
    const uint32_t direct = rp ? 0u : L.direct_rows;                         // 0 or t
    int32_t a0[KTMAX], a1[KTMAX], c0[KTMAX], c1[KTMAX];
#pragma unroll
    for (int i = 0; i < KTMAX; ++i) {
        uint64_t x = 0, y = 0;
        if ((uint32_t)i < k) {                                               // wave-uniform
            const uint64_t a = e0 + i, b = e0 + k + i;
            x = a < L.len ? canon_i64(sp[a], mod.m, mod.mu) : 0;
            y = b < L.len ? canon_i64(sp[b], mod.m, mod.mu) : 0;
        } else if ((uint32_t)i < kt) {
            if (rp) {
                x = in0 ? canon_i64(rp[b0 * t + (i - k)], mod.m, mod.mu) : 0;
                y = in1 ? canon_i64(rp[(b0 + 1) * t + (i - k)], mod.m, mod.mu) : 0;
            } else {
                drbg_pair<ROUNDS>(key, qc, stream, pair, t, (uint32_t)i - k, mod, x, y);
                if (direct) {                                                // systematic share map: draw i - k IS share i - k
                    int64_t* o = op + (size_t)((uint32_t)i - k) * L.out_stride_clerk;
                    if (vec && in1) store2(o, x, y);
                    else {
                        if (in0) o[0] = (int64_t)x;
                        if (in1) o[1] = (int64_t)y;
                    }
                }
            }
        }
        centre_limbs(x, lp, a0[i], a1[i]);
        centre_limbs(y, lp, c0[i], c1[i]);
    }
    for (uint32_t j = direct; j < n; ++j) {
        const uint64_t* row = Mrows + (size_t)(j - direct) * kt;
        const uint64_t a = l31_dot_rt<KTMAX>(row, a0, a1, kt, lp);
        const uint64_t b = l31_dot_rt<KTMAX>(row, c0, c1, kt, lp);
        int64_t* o = op + (size_t)j * L.out_stride_clerk;
        if (vec && in1) store2(o, a, b);
        else {
            if (in0) o[0] = (int64_t)a;
            if (in1) o[1] = (int64_t)b;
        }
    }
}

template <int KTMAX, int ROUNDS>
__global__ __launch_bounds__(kThreads) void packed_gen_l31_rt_kernel(GenLayout L, uint32_t n, uint32_t k, uint32_t t,
                                                                     ModParams mod, L31Params lp, MatArg M, DrbgKey key,
                                                                     uint64_t chunks, uint64_t batches, bool vec) {
    packed_gen_l31_rt_body<KTMAX, ROUNDS>(L, n, k, t, mod, lp, &M.e[0], key, chunks, batches, vec, blockIdx.x);
}

// the same with the matrix in global memory (n (k + t) beyond the kernarg budget, or k + t up to 32): the row entries
// are wave-uniform, so they still arrive by scalar loads
template <int KTMAX, int ROUNDS>
__global__ __launch_bounds__(kThreads) void packed_gen_l31_rtg_kernel(GenLayout L, uint32_t n, uint32_t k, uint32_t t,
                                                                      ModParams mod, L31Params lp,
                                                                      const uint64_t* __restrict__ Mg, DrbgKey key,
                                                                      uint64_t chunks, uint64_t batches, bool vec) {
    packed_gen_l31_rt_body<KTMAX, ROUNDS>(L, n, k, t, mod, lp, Mg, key, chunks, batches, vec, blockIdx.x);
}

// any-shape fallback: one lane = one batch, matrix and randomness read from global memory
__global__ __launch_bounds__(kThreads) void packed_gen_generic_kernel(GenLayout L, uint32_t n, uint32_t k,
                                                                      uint32_t t, ModParams mod, MontParams mont,
                                                                      const uint64_t* __restrict__ Mm,
                                                                      uint64_t chunks, uint64_t batches) {
    uint64_t p, chunk;
    split_item(blockIdx.x, chunks, p, chunk);
    const uint64_t b = chunk * kThreads + threadIdx.x;
    if (b >= batches) return;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand + p * L.rand_stride;
    const uint32_t kt = k + t;
    const uint32_t direct = L.direct_rows;                                   // 0 or t: L.rand then holds the CSPRNG's draws
    for (uint32_t j = 0; j < direct; ++j)
        L.out[p * L.out_stride_participant + (size_t)j * L.out_stride_clerk + b] = (int64_t)canon_i64(rp[b * t + j], mod.m, mod.mu);
    for (uint32_t j = direct; j < n; ++j) {
        U128 acc{0, 0};
        uint32_t since = 0;
        for (uint32_t i = 0; i < kt; ++i) {
            uint64_t v;
            if (i < k) {
                const uint64_t e = b * k + i;
                v = e < L.len ? canon_i64(sp[e], mod.m, mod.mu) : 0;
            } else {
                v = canon_i64(rp[b * t + (i - k)], mod.m, mod.mu);
            }
            mac128(acc, Mm[(size_t)(j - direct) * kt + i], v);
            if (++since == 4) { mont_acc_condsub(acc, mont.p); since = 0; }
        }
        mont_acc_condsub(acc, mont.p);
        L.out[p * L.out_stride_participant + (size_t)j * L.out_stride_clerk + b] =
            (int64_t)mont_redc(acc, mont.p, mont.pinv);
    }
}

template <int ROUNDS>
__global__ __launch_bounds__(kThreads) void drbg_fill_kernel(int64_t* out, size_t stride, size_t batches, uint32_t T,
                                                             uint64_t first_participant, ModParams mod, DrbgKey key,
                                                             uint64_t chunks) {
    uint64_t p, chunk;
    split_item(blockIdx.x, chunks, p, chunk);
    const uint64_t pair = chunk * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair;
    int64_t* o = out + p * stride;
    const QuadCol qc = quad_col(key);
    for (uint32_t i = 0; i < T; ++i) {
        uint64_t r0, r1;
        drbg_pair<ROUNDS>(key, qc, first_participant + p, pair, T, i, mod, r0, r1);
        if (b0 < batches) o[b0 * T + i] = (int64_t)r0;
        if (b0 + 1 < batches) o[(b0 + 1) * T + i] = (int64_t)r1;
    }
}

// =================================================================================================
// K3  clerk combine: exact 128-bit column sums over rows, reduced once at finish.
//     One lane = two adjacent columns; UNROLL independent 16-byte loads in flight.
// =================================================================================================
// acc_add / acc_atomic_add: modarith.hpp

template <bool VEC, int UNROLL>
__device__ __forceinline__ void combine_body(uint64_t* __restrict__ acc_lo, int64_t* __restrict__ acc_hi,
                                             const int64_t* __restrict__ shares, size_t job_stride, size_t n_rows,
                                             size_t row_stride, size_t dimension, size_t rows_per_split, bool atomic,
                                             size_t bx, size_t by, size_t bz) {
    combine_pair<VEC, UNROLL>(acc_lo, acc_hi, shares, job_stride, n_rows, row_stride, dimension, rows_per_split, atomic,
                              bx * kThreads + threadIdx.x, by, bz);                       // clerk_sum.hpp
}

template <bool VEC, int UNROLL>
__global__ __launch_bounds__(kThreads) void combine_update_kernel(uint64_t* __restrict__ acc_lo,
                                                                  int64_t* __restrict__ acc_hi,
                                                                  const int64_t* __restrict__ shares,
                                                                  size_t job_stride, size_t n_rows,
                                                                  size_t row_stride, size_t dimension,
                                                                  size_t rows_per_split, bool atomic) {
    combine_body<VEC, UNROLL>(acc_lo, acc_hi, shares, job_stride, n_rows, row_stride, dimension, rows_per_split, atomic,
                              blockIdx.x, blockIdx.y, blockIdx.z);
}

// The same sum with a FIXED number of workgroups, each walking the (column block, job, row split) items in grid strides:
// launched beside a VALU-bound kernel on another stream it occupies exactly the wave slots and registers it was sized
// for (one workgroup of 4 waves per CU) instead of taking every slot the other kernel's workgroups free.
template <bool VEC, int UNROLL>
__global__ __launch_bounds__(kThreads) void combine_update_walk_kernel(uint64_t* __restrict__ acc_lo, int64_t* __restrict__ acc_hi,
                                                                       const int64_t* __restrict__ shares, size_t job_stride,
                                                                       size_t n_rows, size_t row_stride, size_t dimension,
                                                                       size_t rows_per_split, bool atomic, uint32_t col_blocks,
                                                                       uint32_t jobs, uint64_t items) {
    for (uint64_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint64_t bx = it % col_blocks, rest = it / col_blocks;      // column block fastest: neighbours stream neighbouring lines
        combine_body<VEC, UNROLL>(acc_lo, acc_hi, shares, job_stride, n_rows, row_stride, dimension, rows_per_split, atomic,
                                  bx, rest % jobs, rest / jobs);
    }
}

// =================================================================================================
// Dual-role launch: software pipelining across tiles inside ONE grid.
// Share generation is VALU-bound, the clerk sum HBM-bound; run back to back each leaves the other
// resource idle, and two streams do not mix well (the long-lived clerk-sum workgroups take every wave
// slot the short-lived share-gen workgroups free).  Here one grid carries both kinds of workgroup at a
// fixed ratio - position c * period is the c-th clerk-sum item of tile i (a column block x job x row
// split, accumulated with carry-propagating atomics), every other position a share-gen chunk of tile
// i+1 - so every CU holds a steady mix and the VALU and HBM phases overlap by construction.  The
// shares are still written to HBM by one tile's launch and read back by the next one's (the unfused
// contract of SURVEY.md 8d); only the schedule changes.
// =================================================================================================
// row loads in flight per lane in the clerk-sum role (the share-gen role's 84 VGPRs are allocated anyway; 8 and 24
// measured within 1 %)
static constexpr int kFuseUnroll = 16;
struct FuseArgs {
    uint64_t* acc_lo; int64_t* acc_hi; const int64_t* prev;       // clerk-sum of the previous tile
    size_t job_stride, n_rows, row_stride, dimension, rows_per_split;
    uint32_t col_blocks, jobs, splits;
    uint64_t n_gen, n_comb, period, grid;                          // grid >= n_gen + n_comb
};

__device__ __forceinline__ bool fuse_role(const FuseArgs& F, uint64_t b, uint64_t& idx) {   // true: clerk-sum item
    const uint64_t q = b / F.period, rem = b - q * F.period;
    if (rem == 0 && q < F.n_comb) { idx = q; return true; }
    const uint64_t before = q + (rem ? 1 : 0);                      // clerk-sum positions below b
    idx = b - (before < F.n_comb ? before : F.n_comb);
    return false;
}

__device__ __forceinline__ void fuse_combine(const FuseArgs& F, uint64_t q) {
    const size_t bx = q % F.col_blocks, t = q / F.col_blocks;
    combine_body<true, kFuseUnroll>(F.acc_lo, F.acc_hi, F.prev, F.job_stride, F.n_rows, F.row_stride, F.dimension, F.rows_per_split,
                          F.splits > 1, bx, t % F.jobs, t / F.jobs);
}

// role of workgroup b: true = done (clerk-sum item or idle surplus position), false = share-gen chunk idx
__device__ __forceinline__ bool fuse_dispatch(const FuseArgs& F, uint64_t b, uint64_t& idx) {
    if (fuse_role(F, b, idx)) { fuse_combine(F, idx); return true; }
    return idx >= F.n_gen;
}

template <int K, int T, int ROUNDS>
__global__ SDA_LB(K, T) void fused_packed_l31_kernel(GenLayout L, uint32_t n, ModParams mod, L31Params lp,
                                                                    MatArg M, DrbgKey key, uint64_t chunks,
                                                                    uint64_t batches, FuseArgs F) {
    uint64_t idx;
    if (!fuse_dispatch(F, blockIdx.x, idx)) packed_gen_l31_body<K, T, ROUNDS, true>(L, n, mod, lp, M, key, chunks, batches, idx);
}

template <int KTMAX, int ROUNDS>
__global__ __launch_bounds__(kThreads) void fused_packed_l31_rt_kernel(GenLayout L, uint32_t n, uint32_t k, uint32_t t,
                                                                       ModParams mod, L31Params lp, MatArg M, DrbgKey key,
                                                                       uint64_t chunks, uint64_t batches, FuseArgs F) {
    uint64_t idx;
    if (!fuse_dispatch(F, blockIdx.x, idx))
        packed_gen_l31_rt_body<KTMAX, ROUNDS>(L, n, k, t, mod, lp, &M.e[0], key, chunks, batches, true, idx);
}

template <int K, int T, int ROUNDS>
__global__ __launch_bounds__(kThreads, 3) void fused_packed_mfma_kernel(GenLayout L, uint32_t n, ModParams mod, MontParams mont,
                                                                        const uint64_t* __restrict__ Mbal, DrbgKey key,
                                                                        uint64_t chunks, uint64_t batches, uint32_t iters,
                                                                        uint32_t k_rt, uint32_t t_rt, FuseArgs F) {
    uint64_t idx;
    if (!fuse_dispatch(F, blockIdx.x, idx))
        packed_gen_mfma_body<K, T, ROUNDS>(L, n, mod, mont, Mbal, key, chunks, batches, iters, idx, k_rt, t_rt);
}

template <int ROUNDS>
__global__ __launch_bounds__(kThreads) void fused_additive_kernel(GenLayout L, uint32_t n, ModParams mod, DrbgKey key,
                                                                  uint64_t chunks, FuseArgs F) {
    uint64_t idx;
    if (!fuse_dispatch(F, blockIdx.x, idx)) additive_gen_body<ROUNDS, true>(L, n, mod, key, chunks, idx);
}

#include "narrow_gen.inc.hpp"

__global__ __launch_bounds__(kThreads) void combine_finish_kernel(const uint64_t* __restrict__ acc_lo,
                                                                  const int64_t* __restrict__ acc_hi, size_t count,
                                                                  ModParams mod, int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= count) return;
    out[i] = (int64_t)mod_i128(acc_lo[i], acc_hi[i], mod.m, mod.mu);
}

// =================================================================================================
// K4  packed reconstruct: secrets[k] = R[k x n'] * sums[n'] per batch (R in Montgomery form).
// =================================================================================================
__global__ __launch_bounds__(kThreads) void packed_reconstruct_kernel(const int64_t* __restrict__ shares,
                                                                      size_t row_stride, uint32_t n_rows, uint32_t k,
                                                                      size_t batches, size_t dimension, ModParams mod,
                                                                      MontParams mont,
                                                                      const uint64_t* __restrict__ Rm,
                                                                      int64_t* __restrict__ out, uint32_t e_per_group) {
    // blockIdx.y = a group of e_per_group secrets of every batch: a shape like tss's PSS_155_728_100 (k = 100, 255 clerk rows)
    // has few batches and 25,500 multiply-adds per batch - one thread per batch left the chip at 164 waves (8.5 ms for 1 Mi
    // secrets); the secrets of a batch are independent dot products over the same clerk column
    const size_t b = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= batches) return;
    const uint32_t e0 = blockIdx.y * e_per_group, e1 = e0 + e_per_group < k ? e0 + e_per_group : k;
    for (uint32_t e = e0; e < e1; ++e) {
        const size_t o = b * k + e;
        if (o >= dimension) break;                       // truncate padding (batched.rs:94)
        U128 acc{0, 0};
        uint32_t since = 0;
        for (uint32_t c = 0; c < n_rows; ++c) {
            const uint64_t v = canon_i64(shares[(size_t)c * row_stride + b], mod.m, mod.mu);
            mac128(acc, Rm[(size_t)e * n_rows + c], v);
            if (++since == 4) { mont_acc_condsub(acc, mont.p); since = 0; }
        }
        mont_acc_condsub(acc, mont.p);
        out[o] = (int64_t)mont_redc(acc, mont.p, mont.pinv);
    }
}

// The same for small (k, n'): one lane = two adjacent batches, the n' share pairs fetched ONCE with 16-byte loads and
// kept in registers, the 2k secrets of a workgroup's 512 batches staged in LDS and written as 16-byte coalesced stores
// (the naive form stores with stride k).  R (k x n', Montgomery form) is wave-uniform: scalar loads.
template <int NMAX>
__global__ __launch_bounds__(kThreads) void packed_reconstruct_vec_kernel(const int64_t* __restrict__ shares, size_t row_stride,
                                                                          uint32_t n_rows, uint32_t k, size_t batches,
                                                                          size_t dimension, ModParams mod, MontParams mont,
                                                                          const uint64_t* __restrict__ Rm,
                                                                          int64_t* __restrict__ out) {
    extern __shared__ int64_t stage[];                    // [2 * kThreads][k] = the block's secrets in output order
    const size_t b0 = 2 * ((size_t)blockIdx.x * kThreads + threadIdx.x);
    uint64_t v0[NMAX], v1[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        v0[c] = v1[c] = 0;
        if ((uint32_t)c < n_rows && b0 < batches) {
            if (b0 + 1 < batches) {
                const ll2 v = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(shares + (size_t)c * row_stride + b0));
                v0[c] = canon_i64(v.x, mod.m, mod.mu); v1[c] = canon_i64(v.y, mod.m, mod.mu);
            } else {
                v0[c] = canon_i64(shares[(size_t)c * row_stride + b0], mod.m, mod.mu);
            }
        }
    }
    for (uint32_t e = 0; e < k; ++e) {
        U128 a0{0, 0}, a1{0, 0};
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {
            if ((uint32_t)c < n_rows) {                   // wave-uniform
                const uint64_t r = Rm[(size_t)e * n_rows + c];
                mac128(a0, r, v0[c]); mac128(a1, r, v1[c]);
                if ((c & 3) == 3) { mont_acc_condsub(a0, mont.p); mont_acc_condsub(a1, mont.p); }
            }
        }
        mont_acc_condsub(a0, mont.p); mont_acc_condsub(a1, mont.p);
        stage[(size_t)(2 * threadIdx.x) * k + e] = (int64_t)mont_redc(a0, mont.p, mont.pinv);
        stage[(size_t)(2 * threadIdx.x + 1) * k + e] = (int64_t)mont_redc(a1, mont.p, mont.pinv);
    }
    __syncthreads();
    // the block's 2 * kThreads * k secrets are contiguous in `out`, from element base (a multiple of 2: 16-byte aligned)
    const size_t base = (size_t)blockIdx.x * 2 * kThreads * k;
    const size_t total = (size_t)2 * kThreads * k;
    for (size_t i = 2 * (size_t)threadIdx.x; i < total; i += 2 * kThreads) {
        if (base + i + 1 < dimension) {                   // truncate padding (batched.rs:94)
            ll2 v; v.x = stage[i]; v.y = stage[i + 1];
            *reinterpret_cast<ll2*>(out + base + i) = v;
        } else if (base + i < dimension) {
            out[base + i] = stage[i];
        }
    }
}

// =================================================================================================
// K6  element-wise (a +- b) mod m
// =================================================================================================
__global__ __launch_bounds__(kThreads) void addsub_mod_kernel(const int64_t* __restrict__ a,
                                                              const int64_t* __restrict__ b, size_t len,
                                                              bool subtract, ModParams mod,
                                                              int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= len) return;
    const uint64_t x = canon_i64(a[i], mod.m, mod.mu), y = canon_i64(b[i], mod.m, mod.mu);
    out[i] = (int64_t)(subtract ? submod(x, y, mod.m) : addmod(x, y, mod.m));
}

// participant p = blockIdx.y of the launch's slice: stream = first stream + p, rows at p * stride; one lane = two
// adjacent elements (16-byte accesses when the rows allow it)
template <int ROUNDS>
__global__ __launch_bounds__(kThreads) void full_mask_drbg_kernel(const int64_t* __restrict__ secrets, size_t secrets_stride,
                                                                  size_t len, uint64_t stream, ModParams mod, DrbgKey key,
                                                                  int64_t* __restrict__ mask, size_t mask_stride,
                                                                  int64_t* __restrict__ masked, size_t masked_stride,
                                                                  bool vec) {
    const uint64_t pair = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t b0 = 2 * pair, p = blockIdx.y;
    uint64_t r0, r1;
    drbg_pair<ROUNDS>(key, quad_col(key), stream + p, pair, 1, 0, mod, r0, r1);
    secrets += p * secrets_stride; mask += p * mask_stride; masked += p * masked_stride;
    if (vec && b0 + 1 < len) {
        const ll2 sv = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(secrets + b0));
        ll2 mk, ms;
        mk.x = (int64_t)r0; mk.y = (int64_t)r1;
        ms.x = (int64_t)addmod(canon_i64(sv.x, mod.m, mod.mu), r0, mod.m);
        ms.y = (int64_t)addmod(canon_i64(sv.y, mod.m, mod.mu), r1, mod.m);
        __builtin_nontemporal_store(mk, reinterpret_cast<ll2*>(mask + b0));
        __builtin_nontemporal_store(ms, reinterpret_cast<ll2*>(masked + b0));
        return;
    }
    if (b0 < len) {
        mask[b0] = (int64_t)r0;
        masked[b0] = (int64_t)addmod(canon_i64(secrets[b0], mod.m, mod.mu), r0, mod.m);
    }
    if (b0 + 1 < len) {
        mask[b0 + 1] = (int64_t)r1;
        masked[b0 + 1] = (int64_t)addmod(canon_i64(secrets[b0 + 1], mod.m, mod.mu), r1, mod.m);
    }
}

// =================================================================================================
// K5  rand-0.3 ChaChaRng mask expansion (chacha.rs:36-39, :60-73).
//     Stream of seed s: block j (128-bit counter = j, key = seed words) -> 8 candidates
//     v_m = (word[2m] << 32) | word[2m+1]; candidate accepted iff v < zone; mask = v % m; the i-th
//     mask is the i-th ACCEPTED candidate.  Fast path: assume no rejection (candidate index ==
//     mask index), one lane owns 8 output positions and loops over seeds; a rejected candidate
//     flags its seed, and flagged seeds are corrected by the exact sequential-order slow kernel.
// =================================================================================================
__global__ __launch_bounds__(kThreads) void chacha_mask_fast_kernel(const uint32_t* __restrict__ seeds,
                                                                    size_t n_seeds, size_t dimension, ModParams mod,
                                                                    uint64_t zone, uint64_t* __restrict__ acc_lo,
                                                                    int64_t* __restrict__ acc_hi,
                                                                    RejectRecord* __restrict__ rejects,
                                                                    size_t seeds_per_split) {
    const uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x;   // ChaCha block index
    const size_t pos0 = j * 8;
    if (pos0 >= dimension) return;
    const size_t s_begin = (size_t)blockIdx.y * seeds_per_split;
    size_t s_end = s_begin + seeds_per_split;
    if (s_end > n_seeds) s_end = n_seeds;

    uint64_t lo[8];
    uint32_t hi[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { lo[m] = 0; hi[m] = 0; }

    for (size_t s = s_begin; s < s_end; ++s) {
        uint32_t key[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) key[w] = seeds[s * 8 + w];          // wave-uniform -> SGPRs
        uint32_t o[16];
        chacha_block_lane<20>(key, (uint32_t)j, (uint32_t)(j >> 32), 0u, 0u, o);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint64_t v = ((uint64_t)o[2 * m] << 32) | o[2 * m + 1];
            if (pos0 + m < dimension && v >= zone) {                       // rare: record where, for the correction pass
                const uint32_t k = atomicAdd(&rejects[s].count, 1u);
                if (k < 3) rejects[s].pos[k] = (uint32_t)(pos0 + m);
            }
            // the candidate itself is added, not v % m: the sums are only ever read modulo m (combine_finish), and
            // v == v % m there - one Barrett reduction per column at the end instead of one per mask
            const uint64_t nl = lo[m] + v;
            hi[m] += nl < lo[m] ? 1u : 0u;
            lo[m] = nl;
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
        if (pos0 + m < dimension) acc_atomic_add(acc_lo + pos0 + m, acc_hi + pos0 + m, lo[m], (int64_t)hi[m]);
}

// candidate m (0..7) of a ChaCha block, rand 0.3 next_u64 = (word 2m << 32) | word 2m+1
__device__ __forceinline__ uint64_t block_candidate(const uint32_t (&o)[16], uint32_t m) {
    uint32_t hi = 0, lo = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t)
        if (t == m) { hi = o[2 * t]; lo = o[2 * t + 1]; }
    return ((uint64_t)hi << 32) | lo;
}

// Shift correction for seeds whose <= 3 rejected candidates (all below `dimension`) are recorded: the mask of
// position i is candidate f(i) = i + #{rejected <= f(i)} instead of candidate i.
// where the masks go: into the 128-bit column accumulators (mask combine, chacha.rs:56-77), or - APPLY - onto a
// participant's own secrets, out[s][i] = (secrets[s][i] + mask_i(seed s)) mod m (mask, chacha.rs:36-47)
struct MaskApply {
    const int64_t* secrets; size_t secrets_stride;
    int64_t* out; size_t out_stride;
    ModParams mod;
    __device__ __forceinline__ void put(uint32_t s, uint64_t i, uint64_t candidate) const {
        const uint64_t x = canon_i64(secrets[(size_t)s * secrets_stride + i], mod.m, mod.mu);
        out[(size_t)s * out_stride + i] = (int64_t)addmod(x, barrett_mod64(candidate, mod.m, mod.mu), mod.m);
    }
};

template <bool APPLY>
__global__ __launch_bounds__(kThreads) void chacha_mask_shift_kernel(const uint32_t* __restrict__ seeds,
                                                                     const uint32_t* __restrict__ list,
                                                                     const RejectRecord* __restrict__ rejects,
                                                                     size_t dimension, uint64_t zone,
                                                                     uint64_t* __restrict__ acc_lo,
                                                                     int64_t* __restrict__ acc_hi, MaskApply ap) {
    const uint32_t s = list[blockIdx.y];
    const uint32_t R = rejects[s].count;                                  // 1..3 by construction of the list
    uint32_t x0 = rejects[s].pos[0], x1 = R > 1 ? rejects[s].pos[1] : 0xFFFFFFFFu, x2 = R > 2 ? rejects[s].pos[2] : 0xFFFFFFFFu;
    if (x0 > x1) { const uint32_t t = x0; x0 = x1; x1 = t; }
    if (x1 > x2) { const uint32_t t = x1; x1 = x2; x2 = t; }
    if (x0 > x1) { const uint32_t t = x0; x0 = x1; x1 = t; }
    const uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t i0 = j * 8;
    if (i0 >= dimension || i0 + 7 < x0) return;                           // nothing moves before the first rejection
    uint32_t key[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) key[w] = seeds[(size_t)s * 8 + w];
    uint32_t o0[16], o1[16], oc[16];
    chacha_block_lane<20>(key, (uint32_t)j, (uint32_t)(j >> 32), 0u, 0u, o0);
    chacha_block_lane<20>(key, (uint32_t)(j + 1), (uint32_t)((j + 1) >> 32), 0u, 0u, o1);
    uint64_t cached = ~0ull;                                              // block index held in oc
    for (uint32_t m = 0; m < 8; ++m) {
        const uint64_t i = i0 + m;
        if (i >= dimension) break;
        if (i < x0) continue;
        uint64_t f = i;
#pragma unroll
        for (int it = 0; it < 3; ++it) f = i + (x0 <= f ? 1u : 0u) + (x1 <= f ? 1u : 0u) + (x2 <= f ? 1u : 0u);
        uint64_t nv;
        if (f < dimension) {
            const uint32_t d = (uint32_t)(f - i0);                       // 1 .. 10
            nv = d < 8 ? block_candidate(o0, d) : block_candidate(o1, d - 8);
        } else {
            // dimension - R candidates below `dimension` are accepted; this position is the (t+1)-th accepted one
            // from candidate `dimension` on, and nobody has tested those yet
            uint64_t need = i - (dimension - R) + 1;
            uint64_t idx = dimension;
            for (;;) {
                const uint64_t b = idx >> 3;
                uint64_t v;
                if (b == j) v = block_candidate(o0, (uint32_t)idx & 7u);
                else if (b == j + 1) v = block_candidate(o1, (uint32_t)idx & 7u);
                else {
                    if (b != cached) { chacha_block_lane<20>(key, (uint32_t)b, (uint32_t)(b >> 32), 0u, 0u, oc); cached = b; }
                    v = block_candidate(oc, (uint32_t)idx & 7u);
                }
                if (v < zone && --need == 0) { nv = v; break; }
                ++idx;
            }
        }
        if (APPLY) ap.put(s, i, nv);                                      // overwrite what the fast pass wrote here
        else {
            const uint64_t ov = block_candidate(o0, m);                   // what the fast kernel added here
            acc_atomic_add(acc_lo + i, acc_hi + i, nv - ov, nv < ov ? -1 : 0);
        }
    }
}

// APPLY fast pass: participant s = blockIdx.y of the slice, one lane = one ChaCha block = 8 positions
__global__ __launch_bounds__(kThreads) void chacha_mask_apply_kernel(const uint32_t* __restrict__ seeds, size_t dimension,
                                                                     uint64_t zone, RejectRecord* __restrict__ rejects,
                                                                     MaskApply ap) {
    const uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const size_t pos0 = j * 8;
    if (pos0 >= dimension) return;
    const uint32_t s = blockIdx.y;
    uint32_t key[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) key[w] = seeds[(size_t)s * 8 + w];
    uint32_t o[16];
    chacha_block_lane<20>(key, (uint32_t)j, (uint32_t)(j >> 32), 0u, 0u, o);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (pos0 + m >= dimension) break;
        const uint64_t v = ((uint64_t)o[2 * m] << 32) | o[2 * m + 1];
        if (v >= zone) {
            const uint32_t k = atomicAdd(&rejects[s].count, 1u);
            if (k < 3) rejects[s].pos[k] = (uint32_t)(pos0 + m);
        }
        ap.put(s, pos0 + m, v);
    }
}

// exact expansion for the listed seeds; one workgroup per seed walks the candidate stream in order.
// With subtract_naive the fast kernel's "candidate i -> position i" contribution is taken back.
template <bool APPLY>
__global__ __launch_bounds__(kThreads) void chacha_mask_slow_kernel(const uint32_t* __restrict__ seeds,
                                                                    const uint32_t* __restrict__ list, size_t dimension,
                                                                    ModParams mod, uint64_t zone,
                                                                    uint64_t* __restrict__ acc_lo,
                                                                    int64_t* __restrict__ acc_hi, bool subtract_naive,
                                                                    MaskApply ap) {
    __shared__ uint32_t wave_tot[kThreads / 64];
    __shared__ uint32_t chunk_total;
    const uint32_t s = list ? list[blockIdx.x] : blockIdx.x;
    uint32_t key[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) key[w] = seeds[(size_t)s * 8 + w];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    uint64_t accepted_base = 0;     // masks emitted before this chunk
    uint64_t block_base = 0;        // first ChaCha block index of this chunk
    while (accepted_base < dimension) {
        const uint64_t j = block_base + threadIdx.x;
        uint32_t o[16];
        chacha_block_lane<20>(key, (uint32_t)j, (uint32_t)(j >> 32), 0u, 0u, o);
        uint64_t r[8];
        uint32_t okmask = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint64_t v = ((uint64_t)o[2 * m] << 32) | o[2 * m + 1];
            if (v < zone) okmask |= 1u << m;
            r[m] = barrett_mod64(v, mod.m, mod.mu);
        }
        // exclusive scan of accepted counts over the workgroup (thread order == stream order)
        const uint32_t cnt = __builtin_popcount(okmask);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
        if (threadIdx.x == kThreads - 1) chunk_total = wave_off + incl;
        uint64_t pos = accepted_base + wave_off + (incl - cnt);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!APPLY && subtract_naive) {
                const uint64_t ci = j * 8 + m;                           // naive position of candidate
                if (ci < dimension && r[m] != 0)
                    acc_atomic_add(acc_lo + ci, acc_hi + ci, (uint64_t)0 - r[m], -1);
            }
            if (okmask & (1u << m)) {
                if (pos < dimension) {
                    if (APPLY) ap.put(s, pos, r[m]);
                    else acc_atomic_add(acc_lo + pos, acc_hi + pos, r[m], 0);
                }
                ++pos;
            }
        }
        __syncthreads();
        accepted_base += chunk_total;
        block_base += kThreads;
        __syncthreads();
    }
}

// =================================================================================================
// synthetic bench input: splitmix64(seed ^ (participant << 32 | i)) mod m
// =================================================================================================
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kThreads) void fill_synthetic_kernel(int64_t* out, size_t len, size_t stride,
                                                                  uint64_t first_participant, uint64_t seed,
                                                                  ModParams mod, uint64_t chunks) {
    uint64_t p, chunk;
    split_item(blockIdx.x, chunks, p, chunk);
    const uint64_t i = chunk * kThreads + threadIdx.x;
    if (i >= len) return;
    const uint64_t x = splitmix64(seed ^ (((first_participant + p) << 32) | i));
    out[p * stride + i] = (int64_t)barrett_mod64(x, mod.m, mod.mu);
}

__global__ __launch_bounds__(kThreads) void modsum_parts_kernel(const int64_t* __restrict__ parts, size_t n_parts,
                                                                size_t part_stride, size_t len, ModParams mod,
                                                                int64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= len) return;
    uint64_t lo = 0;
    int64_t hi = 0;
    for (size_t g = 0; g < n_parts; ++g) acc_add(lo, hi, parts[g * part_stride + i]);
    out[i] = (int64_t)mod_i128(lo, hi, mod.m, mod.mu);
}

// =================================================================================================
// launchers
// =================================================================================================
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
// HIP limits a launch to < 2^32 work-items per grid dimension
static constexpr uint64_t kMaxBlocks = (0xFFFFFFFFull / kThreads);
static inline hipError_t grid_check(uint64_t blocks) {
    return blocks > kMaxBlocks ? hipErrorInvalidConfiguration : hipSuccess;
}
// (participant, chunk)-indexed kernels are launched in slices of participants that respect the limit
static inline uint64_t participants_per_launch(uint64_t chunks, uint64_t participants) {
    if (chunks == 0 || chunks > kMaxBlocks) return 0;
    const uint64_t m = kMaxBlocks / chunks;
    return m < participants ? m : participants;
}
static inline GenLayout slice(const GenLayout& L, uint64_t p0, uint64_t count) {
    GenLayout S = L;
    S.secrets = L.secrets + p0 * L.secrets_stride;
    if (L.rand) S.rand = L.rand + p0 * L.rand_stride;
    S.out = L.out + p0 * L.out_stride_participant;
    S.participants = count;
    S.first_participant = L.first_participant + p0;
    return S;
}

static bool gen_vec_ok(const GenLayout& L, size_t secrets_per_lane_even) {
    (void)secrets_per_lane_even;
    return aligned16(L.secrets) && aligned16(L.out) && (L.secrets_stride % 2 == 0) &&
           (L.out_stride_participant % 2 == 0) && (L.out_stride_clerk % 2 == 0);
}

template <int ROUNDS>
static hipError_t additive_launch_r(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key,
                                    hipStream_t s) {
    const uint64_t chunks = ceil_div(ceil_div(L.len, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    const bool vec = gen_vec_ok(L, 0);
    note_kernel("additive_gen_kernel<%d, %s>", ROUNDS, vec ? "true" : "false");
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const unsigned blocks = (unsigned)(chunks * S.participants);
        if (vec) additive_gen_kernel<ROUNDS, true><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, key, chunks);
        else additive_gen_kernel<ROUNDS, false><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, key, chunks);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_additive_generate_signed_drbg(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                                hipStream_t s) {
    if (L.rand || n < 2 || (rounds != 20 && rounds != 12 && rounds != 8)) return hipErrorInvalidValue;
    const uint64_t chunks = ceil_div(ceil_div(L.len, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    note_kernel("signed_additive_gen_drbg_kernel<%d>", rounds);
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const dim3 grid((unsigned)(chunks * S.participants)), block(kThreads);
        if (rounds == 20) signed_additive_gen_drbg_kernel<20><<<grid, block, 0, s>>>(S, n, mod, key, chunks);
        else if (rounds == 12) signed_additive_gen_drbg_kernel<12><<<grid, block, 0, s>>>(S, n, mod, key, chunks);
        else signed_additive_gen_drbg_kernel<8><<<grid, block, 0, s>>>(S, n, mod, key, chunks);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_additive_generate(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key,
                                    int rounds, hipStream_t s) {
    switch (rounds) {
        case 20: return additive_launch_r<20>(L, n, mod, key, s);
        case 12: return additive_launch_r<12>(L, n, mod, key, s);
        case 8: return additive_launch_r<8>(L, n, mod, key, s);
        default: return hipErrorInvalidValue;
    }
}

// compiled (k, t) pairs of the fast packed kernel: the BASELINE shapes and their reference-valid
// (tss FFT) neighbours, plus a few small ones used by tests
#define SDA_PACKED_SHAPES(X) X(3, 1) X(3, 4) X(8, 2) X(8, 7) X(1, 1) X(2, 1) X(1, 2) X(2, 5) X(4, 3)

bool packed_fast_path_available(uint32_t k, uint32_t t, uint32_t n) {
    if ((uint64_t)n * (k + t) > SDA_MAT_ARG_MAX) return false;
#define X(K_, T_) if (k == K_ && t == T_) return true;
    SDA_PACKED_SHAPES(X)
#undef X
    return false;
}

template <int K, int T, int ROUNDS>
static hipError_t packed_launch_kt(const GenLayout& L, uint32_t n, const ModParams& mod, const MontParams& mont,
                                   const MatArg& M, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, K);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    const bool vec = gen_vec_ok(L, 0);
    note_kernel("packed_gen_kernel<%d, %d, %d, %s>", K, T, ROUNDS, vec ? "true" : "false");
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const unsigned blocks = (unsigned)(chunks * S.participants);
        if (vec) packed_gen_kernel<K, T, ROUNDS, true><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, mont, M, key, chunks, batches);
        else packed_gen_kernel<K, T, ROUNDS, false><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, mont, M, key, chunks, batches);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

template <int ROUNDS>
static hipError_t packed_launch_r(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                  const MontParams& mont, const MatArg& M, const DrbgKey& key, hipStream_t s) {
#define X(K_, T_) if (k == K_ && t == T_) return packed_launch_kt<K_, T_, ROUNDS>(L, n, mod, mont, M, key, s);
    SDA_PACKED_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_packed_generate(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                  const MontParams& mont, const MatArg& M, const DrbgKey& key, int rounds,
                                  hipStream_t s) {
    switch (rounds) {
        case 20: return packed_launch_r<20>(L, n, k, t, mod, mont, M, key, s);
        case 12: return packed_launch_r<12>(L, n, k, t, mod, mont, M, key, s);
        case 8: return packed_launch_r<8>(L, n, k, t, mod, mont, M, key, s);
        default: return hipErrorInvalidValue;
    }
}

// every split of k + t = 7 (the tss-valid family for n = 8: k + t + 1 = 2^3), the BASELINE shapes, small shapes and a
// few splits of k + t = 15 (n = 26 / 80); anything else takes the generic kernel (about half the rate)
#define SDA_PACKED_L31_SHAPES(X) X(3, 1) X(3, 4) X(8, 2) X(8, 7) X(1, 1) X(2, 1) X(1, 2) X(2, 2) X(1, 3) X(3, 0) \
    X(4, 0) X(2, 0) X(1, 0) X(2, 5) X(4, 3) X(4, 4) X(5, 3) X(5, 2) X(6, 1) X(1, 6) X(7, 0) X(12, 3) X(10, 5) X(4, 11)

static bool packed_l31_compiled(uint32_t k, uint32_t t) {
#define X(K_, T_) if (k == K_ && t == T_) return true;
    SDA_PACKED_L31_SHAPES(X)
#undef X
    return false;
}
// Montgomery radix of the constants the limb-31 kernels expect for (k, t): 2^93 for the shapes compiled in the three-digit form
// (L31UseR93), 2^62 for everything else (incl. the run-time (k, t) kernels)
unsigned packed_l31_r_bits(uint32_t k, uint32_t t) {
#define X(K_, T_) if (k == K_ && t == T_) return L31UseR93<K_, T_>::value ? 93u : 62u;
    SDA_PACKED_L31_SHAPES(X)
#undef X
    return packed_l31_rt_r_bits(k + t);
}
// the run-time (k, t) kernels (kernarg or global matrix): three digits from 17 terms on (l31_dot_rt)
unsigned packed_l31_rt_r_bits(uint32_t kt) { return kt > 16 ? 93u : 62u; }
// a compiled instance in the three-digit form (the host checks its 8-term / one-group variants on the actual constants)
bool packed_l31_three_digit_compiled(uint32_t k, uint32_t t) { return packed_l31_compiled(k, t) && packed_l31_r_bits(k, t) == 93u; }

bool packed_l31_path_available(uint32_t k, uint32_t t, uint32_t n) {
    if ((uint64_t)n * (k + t) > SDA_MAT_ARG_MAX) return false;
    if (packed_l31_compiled(k, t)) return true;
    return k >= 1 && k + t <= 16 && (uint64_t)n * (k + t) + 6 <= SDA_MAT_ARG_MAX;     // run-time (k, t) kernel: a row is read up to 6 entries past its end
}

template <int K, int T, int ROUNDS>
static hipError_t packed_l31_launch_kt(const GenLayout& L, uint32_t n, const ModParams& mod, const L31Params& lp,
                                       const MatArg& M, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, K);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    const bool vec = gen_vec_ok(L, 0);
    note_kernel("packed_gen_l31_kernel<%d, %d, %d, %s>", K, T, ROUNDS, vec ? "true" : "false");
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const unsigned blocks = (unsigned)(chunks * S.participants);
        if (vec) packed_gen_l31_kernel<K, T, ROUNDS, true><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, lp, M, key, chunks, batches);
        else packed_gen_l31_kernel<K, T, ROUNDS, false><<<dim3(blocks), dim3(kThreads), 0, s>>>(S, n, mod, lp, M, key, chunks, batches);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

template <int KTMAX, int ROUNDS>
static hipError_t packed_l31_launch_rt(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                       const L31Params& lp, const MatArg& M, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const bool vec = aligned16(S.out) && S.out_stride_participant % 2 == 0 && S.out_stride_clerk % 2 == 0;
        note_kernel("packed_gen_l31_rt_kernel<%d, %d>", KTMAX, ROUNDS);
        packed_gen_l31_rt_kernel<KTMAX, ROUNDS><<<dim3((unsigned)(chunks * S.participants)), dim3(kThreads), 0, s>>>(
            S, n, k, t, mod, lp, M, key, chunks, batches, vec);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

template <int ROUNDS>
static hipError_t packed_l31_launch_r(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                      const L31Params& lp, const MatArg& M, const DrbgKey& key, hipStream_t s) {
#define X(K_, T_) if (k == K_ && t == T_) return packed_l31_launch_kt<K_, T_, ROUNDS>(L, n, mod, lp, M, key, s);
    SDA_PACKED_L31_SHAPES(X)
#undef X
    const uint32_t kt = k + t;
    if (kt <= 4) return packed_l31_launch_rt<4, ROUNDS>(L, n, k, t, mod, lp, M, key, s);
    if (kt <= 8) return packed_l31_launch_rt<8, ROUNDS>(L, n, k, t, mod, lp, M, key, s);
    if (kt <= 12) return packed_l31_launch_rt<12, ROUNDS>(L, n, k, t, mod, lp, M, key, s);
    if (kt <= 16) return packed_l31_launch_rt<16, ROUNDS>(L, n, k, t, mod, lp, M, key, s);
    return hipErrorInvalidValue;
}

hipError_t launch_packed_generate_l31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                      const L31Params& lp, const MatArg& M, const DrbgKey& key, int rounds,
                                      hipStream_t s) {
    switch (rounds) {
        case 20: return packed_l31_launch_r<20>(L, n, k, t, mod, lp, M, key, s);
        case 12: return packed_l31_launch_r<12>(L, n, k, t, mod, lp, M, key, s);
        case 8: return packed_l31_launch_r<8>(L, n, k, t, mod, lp, M, key, s);
        default: return hipErrorInvalidValue;
    }
}

template <int KTMAX, int ROUNDS>
static hipError_t packed_l31_launch_rtg(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                        const L31Params& lp, const uint64_t* d_M, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const bool vec = aligned16(S.out) && S.out_stride_participant % 2 == 0 && S.out_stride_clerk % 2 == 0;
        note_kernel("packed_gen_l31_rtg_kernel<%d, %d>", KTMAX, ROUNDS);
        packed_gen_l31_rtg_kernel<KTMAX, ROUNDS><<<dim3((unsigned)(chunks * S.participants)), dim3(kThreads), 0, s>>>(
            S, n, k, t, mod, lp, d_M, key, chunks, batches, vec);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

bool packed_l31_global_path_available(uint32_t k, uint32_t t) { return k >= 1 && k + t <= 64; }

// d_M: n (k + t) limb-31 packed entries followed by seven zero entries (device memory)
hipError_t launch_packed_generate_l31_global(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                             const L31Params& lp, const uint64_t* d_M, const DrbgKey& key, int rounds,
                                             hipStream_t s) {
    const uint32_t kt = k + t;
#define RTG(KTMAX_)                                                                                        \
    (rounds == 20 ? packed_l31_launch_rtg<KTMAX_, 20>(L, n, k, t, mod, lp, d_M, key, s)                    \
     : rounds == 12 ? packed_l31_launch_rtg<KTMAX_, 12>(L, n, k, t, mod, lp, d_M, key, s)                  \
     : rounds == 8 ? packed_l31_launch_rtg<KTMAX_, 8>(L, n, k, t, mod, lp, d_M, key, s) : hipErrorInvalidValue)
    if (kt <= 8) return RTG(8);
    if (kt <= 16) return RTG(16);
    if (kt <= 32) return RTG(32);
    if (kt <= 64) return RTG(64);      // 256 limb registers per lane: one wave per SIMD, still far ahead of the generic kernel
#undef RTG
    return hipErrorInvalidValue;
}

// ---- limb GEMM on the matrix cores (packed_gen_mfma_kernel) ----------------------------------------------------------
#define SDA_MFMA_SHAPES(X) X(8, 7) X(8, 2) X(3, 4) X(3, 1) X(12, 3) X(10, 5) X(4, 11)
static constexpr uint32_t kMfmaIters = 8;                      // 64-batch steps per wave: one workgroup = 2048 batches

static bool packed_mfma_compiled(uint32_t k, uint32_t t) {
#define X(K_, T_) if (k == K_ && t == T_) return true;
    SDA_MFMA_SHAPES(X)
#undef X
    return false;
}
bool packed_mfma_path_available(uint32_t k, uint32_t t, uint32_t n) {
    if (n > (uint32_t)kMfmaMaxClerks) return false;
    return packed_mfma_compiled(k, t) || (k >= 1 && k + t > 8 && k + t <= 16);        // run-time (k, t) form: two MFMA steps
}

template <int K, int T>
static size_t mfma_table_bytes(uint32_t n) { return (size_t)n * (K ? (K + T + 7) / 8 : 2) * 64; }

template <int K, int T, int ROUNDS>
static hipError_t packed_mfma_launch_kt(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                        const MontParams& mont, const uint64_t* d_Mbal, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(batches, (uint64_t)kThreads * kMfmaIters);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        if (mfma_table_bytes<K, T>(n) > 24 * 1024)                      // beyond the default 64 KiB of LDS per workgroup
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&packed_gen_mfma_kernel<K, T, ROUNDS>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_table_bytes<K, T>(n)))
                return e;
        note_kernel("packed_gen_mfma_kernel<%d, %d, %d>", K, T, ROUNDS);
        packed_gen_mfma_kernel<K, T, ROUNDS><<<dim3((unsigned)(chunks * S.participants)), dim3(kThreads), mfma_table_bytes<K, T>(n), s>>>(
            S, n, mod, mont, d_Mbal, key, chunks, batches, kMfmaIters, k, t);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

// d_Mbal: [n][8 * ceil((k + t) / 8)] balanced-byte forms of the Montgomery-form (R = 2^64) matrix, zero padded
hipError_t launch_packed_generate_mfma(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                       const MontParams& mont, const uint64_t* d_Mbal, const DrbgKey& key, int rounds,
                                       hipStream_t s) {
#define X(K_, T_)                                                                                           \
    if (k == K_ && t == T_)                                                                                 \
        return rounds == 20   ? packed_mfma_launch_kt<K_, T_, 20>(L, n, k, t, mod, mont, d_Mbal, key, s)    \
               : rounds == 12 ? packed_mfma_launch_kt<K_, T_, 12>(L, n, k, t, mod, mont, d_Mbal, key, s)    \
               : rounds == 8  ? packed_mfma_launch_kt<K_, T_, 8>(L, n, k, t, mod, mont, d_Mbal, key, s)     \
                              : hipErrorInvalidValue;
    SDA_MFMA_SHAPES(X)
#undef X
    if (!packed_mfma_path_available(k, t, n)) return hipErrorInvalidValue;
    return rounds == 20   ? packed_mfma_launch_kt<0, 0, 20>(L, n, k, t, mod, mont, d_Mbal, key, s)
           : rounds == 12 ? packed_mfma_launch_kt<0, 0, 12>(L, n, k, t, mod, mont, d_Mbal, key, s)
           : rounds == 8  ? packed_mfma_launch_kt<0, 0, 8>(L, n, k, t, mod, mont, d_Mbal, key, s)
                          : hipErrorInvalidValue;
}

hipError_t launch_packed_generate_generic(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t,
                                          const ModParams& mod, const MontParams& mont, const uint64_t* d_Mmont,
                                          hipStream_t s) {
    if (!L.rand && t > 0) return hipErrorInvalidValue;
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(batches, kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        note_kernel("packed_gen_generic_kernel");
        packed_gen_generic_kernel<<<dim3((unsigned)(chunks * S.participants)), dim3(kThreads), 0, s>>>(S, n, k, t, mod, mont,
                                                                                                       d_Mmont, chunks, batches);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

template <int ROUNDS>
static hipError_t drbg_fill_r(int64_t* d_out, size_t stride, size_t participants, size_t batches, uint32_t T,
                              uint64_t first_participant, const ModParams& mod, const DrbgKey& key, hipStream_t s) {
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * participants == 0 || T == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < participants; p0 += per) {
        const uint64_t cnt = per < participants - p0 ? per : participants - p0;
        drbg_fill_kernel<ROUNDS><<<dim3((unsigned)(chunks * cnt)), dim3(kThreads), 0, s>>>(d_out + p0 * stride, stride, batches, T,
                                                                                           first_participant + p0, mod, key, chunks);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_drbg_fill(int64_t* d_out, size_t stride, size_t participants, size_t batches, uint32_t T,
                            uint64_t first_participant, const ModParams& mod, const DrbgKey& key, int rounds,
                            hipStream_t s) {
    switch (rounds) {
        case 20: return drbg_fill_r<20>(d_out, stride, participants, batches, T, first_participant, mod, key, s);
        case 12: return drbg_fill_r<12>(d_out, stride, participants, batches, T, first_participant, mod, key, s);
        case 8: return drbg_fill_r<8>(d_out, stride, participants, batches, T, first_participant, mod, key, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_combine_update(uint64_t* d_acc_lo, int64_t* d_acc_hi, const int64_t* d_shares, size_t jobs,
                                 size_t job_stride, size_t n_rows, size_t row_stride, size_t dimension,
                                 hipStream_t s, unsigned max_wg_per_cu, unsigned walk_workgroups) {
    if (jobs == 0 || n_rows == 0 || dimension == 0) return hipSuccess;
    if (jobs > 65535) return hipErrorInvalidConfiguration;
    const uint64_t col_blocks = ceil_div(ceil_div(dimension, 2), kThreads);
    if (hipError_t e = grid_check(col_blocks)) return e;
    // enough workgroups to fill 256 CUs several times over; split the rows if the columns cannot
    const uint64_t want_blocks = 256 * 16;
    uint64_t split = 1;
    if (col_blocks * jobs < want_blocks) split = ceil_div(want_blocks, col_blocks * jobs);
    const uint64_t max_split = ceil_div(n_rows, 16);            // at least 16 rows per split
    if (split > max_split) split = max_split;
    if (split > 65535) split = 65535;
    if (split < 1) split = 1;
    const size_t rows_per_split = ceil_div(n_rows, split);
    split = ceil_div(n_rows, rows_per_split);
    const bool atomic = split > 1;
    const bool vec = aligned16(d_shares) && (job_stride % 2 == 0) && (row_stride % 2 == 0);
    dim3 grid((unsigned)col_blocks, (unsigned)jobs, (unsigned)split);
    // max_wg_per_cu > 0: an unused dynamic-LDS request caps the resident workgroups per CU, so that this
    // HBM-bound kernel (8 waves per CU already saturate HBM) leaves wave slots to a VALU-bound kernel
    // running on another stream
    const unsigned lds_pad = max_wg_per_cu > 0 ? (160u * 1024u) / max_wg_per_cu - 256u : 0u;
    if (walk_workgroups > 0) {
        const uint64_t items = col_blocks * jobs * split;
        const unsigned wgs = (unsigned)(items < walk_workgroups ? items : walk_workgroups);
        if (vec)
            combine_update_walk_kernel<true, 8><<<dim3(wgs), dim3(kThreads), 0, s>>>(d_acc_lo, d_acc_hi, d_shares, job_stride, n_rows, row_stride,
                                                                                    dimension, rows_per_split, atomic, (uint32_t)col_blocks,
                                                                                    (uint32_t)jobs, items);
        else
            combine_update_walk_kernel<false, 1><<<dim3(wgs), dim3(kThreads), 0, s>>>(d_acc_lo, d_acc_hi, d_shares, job_stride, n_rows, row_stride,
                                                                                     dimension, rows_per_split, atomic, (uint32_t)col_blocks,
                                                                                     (uint32_t)jobs, items);
        return hipGetLastError();
    }
    if (vec)
        combine_update_kernel<true, 8><<<grid, dim3(kThreads), lds_pad, s>>>(d_acc_lo, d_acc_hi, d_shares, job_stride, n_rows,
                                                                       row_stride, dimension, rows_per_split, atomic);
    else
        combine_update_kernel<false, 1><<<grid, dim3(kThreads), 0, s>>>(d_acc_lo, d_acc_hi, d_shares, job_stride, n_rows,
                                                                        row_stride, dimension, rows_per_split, atomic);
    return hipGetLastError();
}

// ---- dual-role launch ----------------------------------------------------------------------------------
static bool fuse_plan(const GenLayout& L, uint64_t chunks, uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev,
                      size_t prev_rows, size_t jobs, size_t dimension, FuseArgs& F) {
    F.acc_lo = acc_lo; F.acc_hi = acc_hi; F.prev = d_prev;
    F.job_stride = L.out_stride_clerk; F.row_stride = L.out_stride_participant;
    F.n_rows = prev_rows; F.dimension = dimension; F.jobs = (uint32_t)jobs;
    F.n_gen = chunks * L.participants;
    F.col_blocks = (uint32_t)ceil_div(ceil_div(dimension, 2), kThreads);
    const bool have_comb = d_prev && prev_rows > 0 && jobs > 0 && dimension > 0;
    // clerk-sum items of up to 512 rows (measured best of 16..2000: 64 and fewer cost 5 % in atomics, one item per
    // column block loses the even mix)
    uint64_t splits = have_comb ? ceil_div(prev_rows, 512) : 1;
    if (splits > 64) splits = 64;
    F.rows_per_split = have_comb ? ceil_div(prev_rows, splits) : 1;
    splits = have_comb ? ceil_div(prev_rows, F.rows_per_split) : 1;
    F.splits = (uint32_t)splits;
    F.n_comb = have_comb ? (uint64_t)F.col_blocks * jobs * splits : 0;
    // XCD-aware role map: workgroup b runs on XCD b % 8, so the period must be odd or the clerk-sum items
    // would pile up on a few XCDs (a period of 16 put ALL of them on XCD 0: 3x slower than serial)
    F.period = F.n_comb ? F.n_gen / F.n_comb + 1 : 1;
    if (F.n_comb && (F.period & 1) == 0) F.period = F.period > 2 ? F.period - 1 : 3;
    F.grid = F.n_gen + F.n_comb;
    if (F.n_comb && (F.n_comb - 1) * F.period + 1 > F.grid) F.grid = (F.n_comb - 1) * F.period + 1;   // surplus positions idle
    if (F.grid == 0 || F.grid > kMaxBlocks) return false;
    if (F.n_gen && !gen_vec_ok(L, 0)) return false;
    if (have_comb && !(aligned16(d_prev) && (F.job_stride % 2 == 0) && (F.row_stride % 2 == 0))) return false;
    return true;
}

template <int K, int T, int ROUNDS>
static hipError_t fused_l31_kt(const GenLayout& L, uint32_t n, const ModParams& mod, const L31Params& lp, const MatArg& M,
                               const DrbgKey& key, const FuseArgs& F, uint64_t chunks, uint64_t batches, hipStream_t s) {
    note_kernel("fused_packed_l31_kernel<%d, %d, %d>", K, T, ROUNDS);
    fused_packed_l31_kernel<K, T, ROUNDS><<<dim3((unsigned)F.grid), dim3(kThreads), 0, s>>>(L, n, mod, lp, M, key,
                                                                                                       chunks, batches, F);
    return hipGetLastError();
}

// the shapes the dual-role launch is compiled for (the BASELINE ones and their tss-valid neighbours)
#define SDA_FUSED_SHAPES(X) X(3, 1) X(3, 4) X(8, 2) X(8, 7)

hipError_t launch_fused_packed_l31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                   const L31Params& lp, const MatArg& M, const DrbgKey& key, int rounds, uint64_t* acc_lo,
                                   int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs, size_t dimension,
                                   hipStream_t s, bool* fused) {
    *fused = false;
    if ((rounds != 20 && rounds != 12 && rounds != 8) || L.rand) return hipSuccess;
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    FuseArgs F;
    if (!fuse_plan(L, chunks, acc_lo, acc_hi, d_prev, prev_rows, jobs, dimension, F)) return hipSuccess;
#define X(K_, T_)                                                                                                        \
    if (k == K_ && t == T_) {                                                                                            \
        *fused = true;                                                                                                   \
        return rounds == 20 ? fused_l31_kt<K_, T_, 20>(L, n, mod, lp, M, key, F, chunks, batches, s)                     \
             : rounds == 12 ? fused_l31_kt<K_, T_, 12>(L, n, mod, lp, M, key, F, chunks, batches, s)                     \
                            : fused_l31_kt<K_, T_, 8>(L, n, mod, lp, M, key, F, chunks, batches, s);                     \
    }
    SDA_FUSED_SHAPES(X)
#undef X
    // no compiled instance: the run-time (k, t) form, when the shape fits it
    const uint32_t kt = k + t;
    if (k < 1 || kt > 16 || (uint64_t)n * kt + 6 > SDA_MAT_ARG_MAX) return hipSuccess;
    *fused = true;
    const dim3 grid((unsigned)F.grid), block(kThreads);
#define RT(KTMAX_)                                                                                                      \
    do {                                                                                                                 \
        note_kernel("fused_packed_l31_rt_kernel<%d, %d>", KTMAX_, rounds);                                               \
        if (rounds == 20) fused_packed_l31_rt_kernel<KTMAX_, 20><<<grid, block, 0, s>>>(L, n, k, t, mod, lp, M, key, chunks, batches, F); \
        else if (rounds == 12) fused_packed_l31_rt_kernel<KTMAX_, 12><<<grid, block, 0, s>>>(L, n, k, t, mod, lp, M, key, chunks, batches, F); \
        else fused_packed_l31_rt_kernel<KTMAX_, 8><<<grid, block, 0, s>>>(L, n, k, t, mod, lp, M, key, chunks, batches, F); \
    } while (0)
    if (kt <= 4) RT(4); else if (kt <= 8) RT(8); else if (kt <= 12) RT(12); else RT(16);
#undef RT
    return hipGetLastError();
}

static constexpr uint32_t kMfmaFusedIters = 2;                 // 512 batches per share-gen item, as in the limb-31 dual-role launch

template <int K, int T, int ROUNDS>
static hipError_t fused_mfma_kt(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod, const MontParams& mont,
                                const uint64_t* d_Mbal, const DrbgKey& key, const FuseArgs& F, uint64_t chunks, uint64_t batches,
                                hipStream_t s) {
    if (mfma_table_bytes<K, T>(n) > 24 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_packed_mfma_kernel<K, T, ROUNDS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_table_bytes<K, T>(n)))
            return e;
    note_kernel("fused_packed_mfma_kernel<%d, %d, %d>", K, T, ROUNDS);
    fused_packed_mfma_kernel<K, T, ROUNDS><<<dim3((unsigned)F.grid), dim3(kThreads), mfma_table_bytes<K, T>(n), s>>>(L, n, mod, mont, d_Mbal, key, chunks, batches,
                                                                                              kMfmaFusedIters, k, t, F);
    return hipGetLastError();
}

hipError_t launch_fused_packed_mfma(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                    const MontParams& mont, const uint64_t* d_Mbal, const DrbgKey& key, int rounds, uint64_t* acc_lo,
                                    int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs, size_t dimension,
                                    hipStream_t s, bool* fused) {
    *fused = false;
    if ((rounds != 20 && rounds != 12 && rounds != 8) || L.rand || !packed_mfma_path_available(k, t, n)) return hipSuccess;
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(batches, (uint64_t)kThreads * kMfmaFusedIters);
    FuseArgs F;
    if (!fuse_plan(L, chunks, acc_lo, acc_hi, d_prev, prev_rows, jobs, dimension, F)) return hipSuccess;
#define X(K_, T_)                                                                                                        \
    if (k == K_ && t == T_) {                                                                                            \
        *fused = true;                                                                                                   \
        return rounds == 20 ? fused_mfma_kt<K_, T_, 20>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s)       \
             : rounds == 12 ? fused_mfma_kt<K_, T_, 12>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s)       \
                            : fused_mfma_kt<K_, T_, 8>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s);       \
    }
    SDA_MFMA_SHAPES(X)
#undef X
    *fused = true;                                                      // run-time (k, t) form
    return rounds == 20 ? fused_mfma_kt<0, 0, 20>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s)
         : rounds == 12 ? fused_mfma_kt<0, 0, 12>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s)
                        : fused_mfma_kt<0, 0, 8>(L, n, k, t, mod, mont, d_Mbal, key, F, chunks, batches, s);
}

// ---- narrow-modulus kernels (narrow_gen.inc.hpp) ---------------------------------------------------------------------
bool packed_n31_path_available(uint32_t k, uint32_t t, uint32_t rows, uint64_t p) {
    return p < (1ull << 31) && k >= 1 && k + t <= 16 && (uint64_t)rows * (k + t) + 3 <= 2 * SDA_MAT_ARG_MAX;   // a row is read up to 3 entries past its end
}

#define SDA_N31_DISPATCH(KERNEL, GRID, ...)                                                                              \
    do {                                                                                                                 \
        const uint32_t kt_ = k + t;                                                                                      \
        const bool g16_ = np.p < (1u << 29);        /* GROUP * p < 2^33: 16 terms per reduction below 2^29, else 4 */    \
        const dim3 grid_((unsigned)(GRID)), block_(kThreads);                                                            \
        note_kernel(#KERNEL "<%u, %d, 20>", kt_ <= 4 ? 4u : kt_ <= 8 ? 8u : kt_ <= 12 ? 12u : 16u, kt_ <= 4 ? 4 : g16_ ? 16 : 4); \
        if (kt_ <= 4) KERNEL<4, 4, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__);                                            \
        else if (kt_ <= 8) { if (g16_) KERNEL<8, 16, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); else KERNEL<8, 4, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); }   \
        else if (kt_ <= 12) { if (g16_) KERNEL<12, 16, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); else KERNEL<12, 4, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); } \
        else { if (g16_) KERNEL<16, 16, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); else KERNEL<16, 4, 20><<<grid_, block_, 0, s>>>(__VA_ARGS__); }               \
    } while (0)

hipError_t launch_packed_generate_n31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                      const N31Params& np, const MatArg& M, const DrbgKey& key, hipStream_t s) {
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    if (chunks * L.participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, L.participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < L.participants; p0 += per) {
        const GenLayout S = slice(L, p0, per < L.participants - p0 ? per : L.participants - p0);
        const bool vec = aligned16(S.out) && S.out_stride_participant % 2 == 0 && S.out_stride_clerk % 2 == 0;
        SDA_N31_DISPATCH(packed_gen_n31_kernel, chunks * S.participants, S, n, k, t, mod, np, M, key, chunks, batches, vec);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_fused_packed_n31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod, const N31Params& np,
                                   const MatArg& M, const DrbgKey& key, uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev,
                                   size_t prev_rows, size_t jobs, size_t dimension, hipStream_t s, bool* fused) {
    *fused = false;
    if (L.rand) return hipSuccess;
    const uint64_t batches = ceil_div(L.len, k);
    const uint64_t chunks = ceil_div(ceil_div(batches, 2), kThreads);
    FuseArgs F;
    if (!fuse_plan(L, chunks, acc_lo, acc_hi, d_prev, prev_rows, jobs, dimension, F)) return hipSuccess;
    *fused = true;
    SDA_N31_DISPATCH(fused_packed_n31_kernel, F.grid, L, n, k, t, mod, np, M, key, chunks, batches, F);
    return hipGetLastError();
}
#undef SDA_N31_DISPATCH

hipError_t launch_fused_additive(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                 uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs,
                                 size_t dimension, hipStream_t s, bool* fused) {
    *fused = false;
    if ((rounds != 20 && rounds != 12 && rounds != 8) || L.rand) return hipSuccess;
    const uint64_t chunks = ceil_div(ceil_div(L.len, 2), kThreads);
    FuseArgs F;
    if (!fuse_plan(L, chunks, acc_lo, acc_hi, d_prev, prev_rows, jobs, dimension, F)) return hipSuccess;
    *fused = true;
    const dim3 grid((unsigned)F.grid), block(kThreads);
    note_kernel("fused_additive_kernel<%d>", rounds);
    if (rounds == 20) fused_additive_kernel<20><<<grid, block, 0, s>>>(L, n, mod, key, chunks, F);
    else if (rounds == 12) fused_additive_kernel<12><<<grid, block, 0, s>>>(L, n, mod, key, chunks, F);
    else fused_additive_kernel<8><<<grid, block, 0, s>>>(L, n, mod, key, chunks, F);
    return hipGetLastError();
}

hipError_t launch_combine_finish(const uint64_t* d_acc_lo, const int64_t* d_acc_hi, size_t count, const ModParams& mod,
                                 int64_t* d_out, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const uint64_t blocks = ceil_div(count, kThreads);
    if (hipError_t e = grid_check(blocks)) return e;
    combine_finish_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, s>>>(d_acc_lo, d_acc_hi, count, mod, d_out);
    return hipGetLastError();
}

// narrow primes (p < 2^31), small shapes and vectorisable layouts only; false: the caller takes launch_packed_reconstruct
bool packed_reconstruct_n31_available(uint32_t n_rows, uint32_t k, uint64_t p, const int64_t* d_shares, size_t row_stride,
                                      const int64_t* d_out) {
    return p < (1ull << 31) && n_rows <= 16 && k <= 16 && aligned16(d_shares) && aligned16(d_out) && row_stride % 2 == 0;
}
hipError_t launch_packed_reconstruct_n31(const int64_t* d_shares, size_t row_stride, uint32_t n_rows, uint32_t k, size_t batches,
                                         size_t dimension, const ModParams& mod, const N31Params& np, const int32_t* d_R31,
                                         int64_t* d_out, hipStream_t s) {
    if (batches == 0) return hipSuccess;
    const uint64_t vblocks = ceil_div(ceil_div(batches, 2), kThreads);
    if (hipError_t e = grid_check(vblocks)) return e;
    const size_t lds = (size_t)2 * kThreads * k * 8;
    const dim3 grid((unsigned)vblocks), block(kThreads);
    const bool g16 = np.p < (1u << 29);
#define RN(NMAX_)                                                                                                                  \
    do {                                                                                                                           \
        if (g16) packed_reconstruct_n31_kernel<NMAX_, 16><<<grid, block, lds, s>>>(d_shares, row_stride, n_rows, k, batches, dimension, mod, np, d_R31, d_out); \
        else packed_reconstruct_n31_kernel<NMAX_, 4><<<grid, block, lds, s>>>(d_shares, row_stride, n_rows, k, batches, dimension, mod, np, d_R31, d_out);      \
    } while (0)
    if (n_rows <= 4) RN(4); else if (n_rows <= 8) RN(8); else RN(16);
#undef RN
    return hipGetLastError();
}

hipError_t launch_packed_reconstruct(const int64_t* d_shares, size_t row_stride, uint32_t n_rows, uint32_t k,
                                     size_t batches, size_t dimension, const ModParams& mod, const MontParams& mont,
                                     const uint64_t* d_Rmont, int64_t* d_out, hipStream_t s) {
    if (batches == 0) return hipSuccess;
    // small shapes (the BASELINE ones): vector loads, registers, LDS-staged coalesced stores
    const bool vec = n_rows <= 16 && k <= 16 && aligned16(d_shares) && aligned16(d_out) && row_stride % 2 == 0;
    if (vec) {
        const uint64_t vblocks = ceil_div(ceil_div(batches, 2), kThreads);
        if (hipError_t e = grid_check(vblocks)) return e;
        const size_t lds = (size_t)2 * kThreads * k * 8;
        if (n_rows <= 4)
            packed_reconstruct_vec_kernel<4><<<dim3((unsigned)vblocks), dim3(kThreads), lds, s>>>(d_shares, row_stride, n_rows, k, batches, dimension, mod, mont, d_Rmont, d_out);
        else if (n_rows <= 8)
            packed_reconstruct_vec_kernel<8><<<dim3((unsigned)vblocks), dim3(kThreads), lds, s>>>(d_shares, row_stride, n_rows, k, batches, dimension, mod, mont, d_Rmont, d_out);
        else
            packed_reconstruct_vec_kernel<16><<<dim3((unsigned)vblocks), dim3(kThreads), lds, s>>>(d_shares, row_stride, n_rows, k, batches, dimension, mod, mont, d_Rmont, d_out);
        return hipGetLastError();
    }
    const uint64_t blocks = ceil_div(batches, kThreads);
    if (hipError_t e = grid_check(blocks)) return e;
    // enough workgroups for the chip: the k secrets of a batch are split into groups when the batches alone do not fill it
    uint64_t groups = blocks < 2048 ? ceil_div(2048, blocks) : 1;
    if (groups > k) groups = k;
    const uint32_t e_per_group = (uint32_t)ceil_div(k, groups);
    groups = ceil_div(k, e_per_group);
    packed_reconstruct_kernel<<<dim3((unsigned)blocks, (unsigned)groups), dim3(kThreads), 0, s>>>(d_shares, row_stride, n_rows, k, batches,
                                                                                                 dimension, mod, mont, d_Rmont, d_out, e_per_group);
    return hipGetLastError();
}

hipError_t launch_addsub_mod(const int64_t* d_a, const int64_t* d_b, size_t len, bool subtract, const ModParams& mod,
                             int64_t* d_out, hipStream_t s) {
    if (len == 0) return hipSuccess;
    const uint64_t blocks = ceil_div(len, kThreads);
    if (hipError_t e = grid_check(blocks)) return e;
    addsub_mod_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, s>>>(d_a, d_b, len, subtract, mod, d_out);
    return hipGetLastError();
}

hipError_t launch_full_mask_drbg(const int64_t* d_secrets, size_t secrets_stride, size_t participants, size_t len,
                                 uint64_t first_stream, const ModParams& mod, const DrbgKey& key, int rounds,
                                 int64_t* d_mask, size_t mask_stride, int64_t* d_masked, size_t masked_stride, hipStream_t s) {
    if (len == 0 || participants == 0) return hipSuccess;
    const uint64_t blocks = ceil_div(ceil_div(len, 2), kThreads);
    if (hipError_t e = grid_check(blocks)) return e;
    const bool vec = aligned16(d_secrets) && aligned16(d_mask) && aligned16(d_masked) && secrets_stride % 2 == 0 &&
                     mask_stride % 2 == 0 && masked_stride % 2 == 0;
    // < 2^32 work-items per launch and <= 65535 in y: slices of participants
    uint64_t per = 0xFFFFFFFFull / (blocks * kThreads);
    if (per > 65535) per = 65535;
    if (per == 0) return hipErrorInvalidConfiguration;
    for (size_t p0 = 0; p0 < participants; p0 += per) {
        const unsigned np = (unsigned)(participants - p0 < per ? participants - p0 : per);
        const dim3 grid((unsigned)blocks, np);
        const int64_t* sp = d_secrets + p0 * secrets_stride;
        int64_t* mp = d_mask + p0 * mask_stride;
        int64_t* xp = d_masked + p0 * masked_stride;
        switch (rounds) {
            case 20: full_mask_drbg_kernel<20><<<grid, dim3(kThreads), 0, s>>>(sp, secrets_stride, len, first_stream + p0, mod, key, mp, mask_stride, xp, masked_stride, vec); break;
            case 12: full_mask_drbg_kernel<12><<<grid, dim3(kThreads), 0, s>>>(sp, secrets_stride, len, first_stream + p0, mod, key, mp, mask_stride, xp, masked_stride, vec); break;
            case 8: full_mask_drbg_kernel<8><<<grid, dim3(kThreads), 0, s>>>(sp, secrets_stride, len, first_stream + p0, mod, key, mp, mask_stride, xp, masked_stride, vec); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

hipError_t launch_chacha_mask_accumulate(const uint32_t* d_seeds, size_t n_seeds, size_t dimension,
                                         const ModParams& mod, uint64_t zone, uint64_t* d_acc_lo, int64_t* d_acc_hi,
                                         RejectRecord* d_rejects, hipStream_t s) {
    if (n_seeds == 0 || dimension == 0) return hipSuccess;
    const uint64_t pos_blocks = ceil_div(ceil_div(dimension, 8), kThreads);
    if (hipError_t e = grid_check(pos_blocks)) return e;
    const uint64_t want_blocks = 256 * 8;
    uint64_t split = 1;
    if (pos_blocks < want_blocks) split = ceil_div(want_blocks, pos_blocks);
    if (split > n_seeds) split = n_seeds;
    if (split > 65535) split = 65535;
    const size_t per = ceil_div(n_seeds, split);
    split = ceil_div(n_seeds, per);
    chacha_mask_fast_kernel<<<dim3((unsigned)pos_blocks, (unsigned)split), dim3(kThreads), 0, s>>>(
        d_seeds, n_seeds, dimension, mod, zone, d_acc_lo, d_acc_hi, d_rejects, per);
    return hipGetLastError();
}

hipError_t launch_chacha_mask_shift(const uint32_t* d_seeds, const uint32_t* d_list, size_t n_list,
                                    const RejectRecord* d_rejects, size_t dimension, const ModParams& mod, uint64_t zone,
                                    uint64_t* d_acc_lo, int64_t* d_acc_hi, hipStream_t s) {
    (void)mod;
    if (n_list == 0 || dimension == 0) return hipSuccess;
    const uint64_t pos_blocks = ceil_div(ceil_div(dimension, 8), kThreads);
    if (hipError_t e = grid_check(pos_blocks)) return e;
    for (size_t l0 = 0; l0 < n_list; l0 += 65535) {
        const unsigned nl = (unsigned)(n_list - l0 < 65535 ? n_list - l0 : 65535);
        chacha_mask_shift_kernel<false><<<dim3((unsigned)pos_blocks, nl), dim3(kThreads), 0, s>>>(
            d_seeds, d_list + l0, d_rejects, dimension, zone, d_acc_lo, d_acc_hi, MaskApply{});
    }
    return hipGetLastError();
}

hipError_t launch_chacha_mask_slow(const uint32_t* d_seeds, const uint32_t* d_list, size_t n_list, size_t dimension,
                                   const ModParams& mod, uint64_t zone, uint64_t* d_acc_lo, int64_t* d_acc_hi,
                                   bool subtract_naive, hipStream_t s) {
    if (n_list == 0 || dimension == 0) return hipSuccess;
    if (hipError_t e = grid_check(n_list)) return e;
    chacha_mask_slow_kernel<false><<<dim3((unsigned)n_list), dim3(kThreads), 0, s>>>(d_seeds, d_list, dimension, mod, zone,
                                                                                     d_acc_lo, d_acc_hi, subtract_naive, MaskApply{});
    return hipGetLastError();
}

// ---- masks applied to each participant's own vector (chacha.rs:24-54 for a device-resident tile) ------------------
hipError_t launch_chacha_apply_fast(const uint32_t* d_seeds, size_t participants, size_t dimension, const ModParams& mod,
                                    uint64_t zone, const int64_t* d_secrets, size_t secrets_stride, int64_t* d_out,
                                    size_t out_stride, RejectRecord* d_rejects, hipStream_t s) {
    if (participants == 0 || dimension == 0) return hipSuccess;
    const uint64_t pos_blocks = ceil_div(ceil_div(dimension, 8), kThreads);
    if (hipError_t e = grid_check(pos_blocks)) return e;
    uint64_t per = 0xFFFFFFFFull / (pos_blocks * kThreads);
    if (per > 65535) per = 65535;
    if (per == 0) return hipErrorInvalidConfiguration;
    for (size_t p0 = 0; p0 < participants; p0 += per) {
        const unsigned np = (unsigned)(participants - p0 < per ? participants - p0 : per);
        const MaskApply ap{d_secrets + p0 * secrets_stride, secrets_stride, d_out + p0 * out_stride, out_stride, mod};
        chacha_mask_apply_kernel<<<dim3((unsigned)pos_blocks, np), dim3(kThreads), 0, s>>>(d_seeds + p0 * 8, dimension, zone,
                                                                                         d_rejects + p0, ap);
    }
    return hipGetLastError();
}

hipError_t launch_chacha_apply_repair(const uint32_t* d_seeds, const uint32_t* d_shift_list, size_t n_shift,
                                      const uint32_t* d_exact_list, size_t n_exact, const RejectRecord* d_rejects,
                                      size_t dimension, const ModParams& mod, uint64_t zone, const int64_t* d_secrets,
                                      size_t secrets_stride, int64_t* d_out, size_t out_stride, hipStream_t s) {
    if (dimension == 0) return hipSuccess;
    const MaskApply ap{d_secrets, secrets_stride, d_out, out_stride, mod};
    const uint64_t pos_blocks = ceil_div(ceil_div(dimension, 8), kThreads);
    if (hipError_t e = grid_check(pos_blocks)) return e;
    for (size_t l0 = 0; l0 < n_shift; l0 += 65535) {
        const unsigned nl = (unsigned)(n_shift - l0 < 65535 ? n_shift - l0 : 65535);
        chacha_mask_shift_kernel<true><<<dim3((unsigned)pos_blocks, nl), dim3(kThreads), 0, s>>>(
            d_seeds, d_shift_list + l0, d_rejects, dimension, zone, nullptr, nullptr, ap);
    }
    if (n_exact) {      // d_exact_list == nullptr: every participant 0..n_exact-1 in stream order
        if (hipError_t e = grid_check(n_exact)) return e;
        chacha_mask_slow_kernel<true><<<dim3((unsigned)n_exact), dim3(kThreads), 0, s>>>(d_seeds, d_exact_list, dimension, mod,
                                                                                        zone, nullptr, nullptr, false, ap);
    }
    return hipGetLastError();
}

hipError_t launch_modsum_parts(const int64_t* d_parts, size_t parts, size_t part_stride, size_t len,
                               const ModParams& mod, int64_t* d_out, hipStream_t s) {
    if (len == 0) return hipSuccess;
    const uint64_t blocks = ceil_div(len, kThreads);
    if (hipError_t e = grid_check(blocks)) return e;
    modsum_parts_kernel<<<dim3((unsigned)blocks), dim3(kThreads), 0, s>>>(d_parts, parts, part_stride, len, mod, d_out);
    return hipGetLastError();
}

hipError_t launch_fill_synthetic(int64_t* d_out, size_t participants, size_t len, size_t stride,
                                 uint64_t first_participant, uint64_t seed, const ModParams& mod, hipStream_t s) {
    const uint64_t chunks = ceil_div(len, kThreads);
    if (chunks * participants == 0) return hipSuccess;
    const uint64_t per = participants_per_launch(chunks, participants);
    if (per == 0) return hipErrorInvalidConfiguration;
    for (uint64_t p0 = 0; p0 < participants; p0 += per) {
        const uint64_t cnt = per < participants - p0 ? per : participants - p0;
        fill_synthetic_kernel<<<dim3((unsigned)(chunks * cnt)), dim3(kThreads), 0, s>>>(d_out + p0 * stride, len, stride,
                                                                                        first_participant + p0, seed, mod, chunks);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

}  // namespace sda
