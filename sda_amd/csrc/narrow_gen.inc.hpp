// Narrow-modulus packed-Shamir share generation (p < 2^31) - included by sda_kernels.hip inside namespace sda, after the
// dual-role helpers (it uses drbg_pair, store2, split_item, FuseArgs / fuse_dispatch of that file).
//
// Why: the reference's packed path multiplies i64 residues WITHOUT widening (tss 0.2; the note at additive.rs:37-39 "we
// assume that the values are really i32"), so its whole valid domain is p^2 + p < 2^63 - p = 433 (full_loop.rs:57-64),
// tss's shipped 746497 and 5038849.  The 62-bit kernels pay two-limb arithmetic for those primes too: four v_mad_i64_i32
// per term and a ~20-instruction radix-2^31 reduction per five terms.  Here a residue is ONE signed 32-bit limb:
//   * values and Montgomery-form constants (R = 2^32) are centred to [-(p-1)/2, (p-1)/2], |.| < 2^30;
//   * a dot product is ONE v_mad_i64_i32 per term into a signed 64-bit sum, GROUP terms at a time with
//     GROUP * p < 2^33 (4 terms for any p < 2^31, 16 below 2^29 - a whole (8,7) row);
//   * Montgomery reduction of the sum S is three 32-bit instructions: q = lo(S) * (-p^-1) (signed), and the exact quotient
//     (S + q p) / 2^32 = hi(S) + mulhi(q, p) + (lo(S) != 0) - the low words cancel, so the carry is known without
//     forming the 65-bit sum; |result| < GROUP p^2 / 2^34 + p / 2 < p, made canonical with one masked add.
// Bounds and exactness: tests/test_narrow_model.py (big-int model, extreme operands).  Draws are the same sda-drbg-v1
// stream (64-bit Lemire sampling), so one CPU restatement of the stream serves both widths; only ROUNDS = 20 is instantiated
// (the other round counts exist for A/B runs of the wide kernels).
//
// k and t are kernel arguments (KTMAX = 4 / 8 / 12 / 16 bounds k + t); the matrix travels in the kernarg segment as
// int32 constants (MatArg reinterpreted: 896 entries), rows back to back, zero padded by three entries.

// sda-drbg-v1 draws for a modulus below 2^32: the same words, the same acceptance rule (lo64(x m) >= 2^64 mod m) and value
// (hi64(x m)) as drbg_pair, with the 96-bit product x m formed from two 32 x 32 -> 64 multiply-adds instead of a 64 x 64 one
__device__ __forceinline__ bool lemire_sample_m32(uint64_t x, uint32_t m, uint64_t thr, uint64_t& out) {
    const uint64_t t0 = (uint64_t)(uint32_t)x * m;
    const uint64_t t1 = (uint64_t)(uint32_t)(x >> 32) * m + (t0 >> 32);
    out = t1 >> 32;
    return ((t1 << 32) | (uint32_t)t0) >= thr;
}
template <int ROUNDS>
__device__ __forceinline__ void drbg_pair_m32(const DrbgKey& key, const QuadCol& qc, uint64_t stream, uint64_t pair, uint32_t T,
                                              uint32_t i, const ModParams& mod, uint64_t& r0, uint64_t& r1) {
    if (drbg_paired(mod.m)) {                               // uniform: the paired rule (modarith.hpp) lives in drbg_pair
        drbg_pair<ROUNDS>(key, qc, stream, pair, T, i, mod, r0, r1);
        return;
    }
    const uint32_t c = threadIdx.x & 3;
    const uint64_t I = (pair >> 2) * (uint64_t)T + i;
    const uint32_t ctr = c == 0 ? (uint32_t)I : c == 1 ? (uint32_t)(I >> 32) : c == 2 ? (uint32_t)stream
                                                                              : ((uint32_t)(stream >> 32) & 0xFFFFFFu);
    uint32_t o0, o1, o2, o3;
    chacha_block_quad<ROUNDS>(qc.cst, qc.kb, qc.kc, ctr, o0, o1, o2, o3);
    const bool ok0 = lemire_sample_m32(((uint64_t)o0 << 32) | o1, (uint32_t)mod.m, mod.lemire_thr, r0);
    const bool ok1 = lemire_sample_m32(((uint64_t)o2 << 32) | o3, (uint32_t)mod.m, mod.lemire_thr, r1);
    if (__builtin_expect(!ok0, 0))
        r0 = drbg_retry<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                (2 * pair) * (uint64_t)T + i, mod.m, mod.lemire_thr);
    if (__builtin_expect(!ok1, 0))
        r1 = drbg_retry<ROUNDS>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                (2 * pair + 1) * (uint64_t)T + i, mod.m, mod.lemire_thr);
}

__device__ __forceinline__ int32_t n31_centre(uint64_t v, const N31Params& P) {          // canonical [0, p) -> centred
    const uint32_t x = (uint32_t)v;
    return (int32_t)(x >= P.h ? x - P.p : x);
}

// S (|S| < 2^62, any exact multiple structure) -> S * 2^-32 mod p, canonical [0, p)
__device__ __forceinline__ uint32_t n31_redc(int64_t S, const N31Params& P) {
    const uint32_t sl = (uint32_t)S;
    const int32_t sh = (int32_t)(S >> 32);
    const int32_t q = (int32_t)(sl * P.pinv);
    int32_t t = sh + __mulhi(q, (int32_t)P.p) + (sl != 0 ? 1 : 0);                       // in (-p, p)
    t += (t >> 31) & (int32_t)P.p;
    return (uint32_t)t;
}

// The dot products of ONE matrix row with the lane's two batches: sum_i m[i] * x[i] and sum_i m[i] * y[i] mod p, canonical.
// Straight-line code: the launchers pick KTMAX as k + t rounded up to a multiple of four and the value limbs beyond k + t
// are zero, so every term is multiplied unconditionally (<= 3 wasted per dot product) - no wave-uniform branches, the two
// accumulator chains alternate (two independent v_mad_i64_i32 in flight instead of one serial chain), and the row's
// constants are SGPR operands of one scalar fetch per row.  (Round 4's first version guarded every group of four terms with a
// branch and fetched its four constants right before use; this form is 1 - 2 % faster on (8,7,26) - 83.5 -> 84.4 Gelem/s -
// which shows that the kernel waits for its ChaCha20 chains, not for its dot products.)
template <int KTMAX, int GROUP>
__device__ __forceinline__ void n31_dot2(const int32_t (&m)[KTMAX], const int32_t (&x)[KTMAX], const int32_t (&y)[KTMAX],
                                         const N31Params& P, uint32_t& ra, uint32_t& rb) {
    uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
    for (int g0 = 0; g0 < KTMAX; g0 += GROUP) {
        int64_t Sa = 0, Sb = 0;
#pragma unroll
        for (int i = g0; i < g0 + GROUP && i < KTMAX; ++i) {
            Sa = i == g0 ? mul_sv(m[i], x[i]) : mad_sv(m[i], x[i], Sa);
            Sb = i == g0 ? mul_sv(m[i], y[i]) : mad_sv(m[i], y[i], Sb);
        }
        const uint32_t ta = n31_redc(Sa, P), tb = n31_redc(Sb, P);
        if (g0 == 0) { acc_a = ta; acc_b = tb; }
        else {
            uint32_t s = acc_a + ta, d = s - P.p;                            // s < 2p < 2^32; s - p wraps exactly when s < p
            acc_a = d < s ? d : s;
            s = acc_b + tb; d = s - P.p;
            acc_b = d < s ? d : s;
        }
    }
    ra = acc_a; rb = acc_b;
}

template <int KTMAX, int GROUP, int ROUNDS>
__device__ __forceinline__ void packed_gen_n31_body(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                                    const N31Params& np, const int32_t* __restrict__ Mrows, const DrbgKey& key,
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
    const uint32_t direct = rp ? 0u : L.direct_rows;                         // 0 or t (systematic share map)
    int32_t x[KTMAX], y[KTMAX];
#pragma unroll
    for (int i = 0; i < KTMAX; ++i) {
        uint64_t a = 0, b = 0;
        if ((uint32_t)i < k) {                                               // wave-uniform
            const uint64_t ea = e0 + i, eb = e0 + k + i;
            a = ea < L.len ? canon_i64(sp[ea], mod.m, mod.mu) : 0;           // zero padding (batched.rs:37-43)
            b = eb < L.len ? canon_i64(sp[eb], mod.m, mod.mu) : 0;
        } else if ((uint32_t)i < kt) {
            if (rp) {
                a = in0 ? canon_i64(rp[b0 * t + (i - k)], mod.m, mod.mu) : 0;
                b = in1 ? canon_i64(rp[(b0 + 1) * t + (i - k)], mod.m, mod.mu) : 0;
            } else {
                drbg_pair_m32<ROUNDS>(key, qc, stream, pair, t, (uint32_t)i - k, mod, a, b);
                if (direct) {                                                // draw i - k IS share i - k
                    int64_t* o = op + (size_t)((uint32_t)i - k) * L.out_stride_clerk;
                    if (vec && in1) store2(o, a, b);
                    else {
                        if (in0) o[0] = (int64_t)a;
                        if (in1) o[1] = (int64_t)b;
                    }
                }
            }
        }
        x[i] = n31_centre(a, np);
        y[i] = n31_centre(b, np);
    }
    // the whole row's constants in ONE scalar fetch at the top of the iteration (they are SGPR operands of the multiply-adds);
    // double-buffering them one row ahead was tried: 32 more SGPRs live, spilled to VGPR lanes, no gain over the other
    // waves of the SIMD hiding this one latency per row.  Indexed from Mrows every time: a pointer carried round the loop
    // loses the kernarg address space and hipcc then copies the whole matrix to scratch.
    for (uint32_t j = direct; j < n; ++j) {
        int32_t m[KTMAX];
#pragma unroll
        for (int i = 0; i < KTMAX; ++i) m[i] = Mrows[(size_t)(j - direct) * kt + i];
        uint32_t a, b;
        n31_dot2<KTMAX, GROUP>(m, x, y, np, a, b);
        int64_t* o = op + (size_t)j * L.out_stride_clerk;
        if (vec && in1) store2(o, a, b);
        else {
            if (in0) o[0] = (int64_t)a;
            if (in1) o[1] = (int64_t)b;
        }
    }
}

// Register caps as for the limb-31 kernels (SDA_LB): hipcc's free allocation takes 117 VGPRs (KTMAX = 8) to 151 (16) - it
// hoists all 2 x KTMAX secret loads to the top - i.e. 4 and 3 waves per SIMD, and the round-4 counters showed these kernels
// waiting, not computing (narrow26_ref: VALU busy 0.53 at 4.6 wave-instructions per element, 0.71 of the HBM floor): the
// serial ChaCha20 chains of the t draws need more waves to hide behind.  5 waves (96 VGPRs) up to 8 terms, 4 (128) beyond.
#ifndef SDA_N31_W_SMALL
#define SDA_N31_W_SMALL 5
#endif
#ifndef SDA_N31_W_BIG
#define SDA_N31_W_BIG 4
#endif
#define SDA_N31_LB(KTMAX_) __launch_bounds__(kThreads, ((KTMAX_) <= 8 ? SDA_N31_W_SMALL : SDA_N31_W_BIG))
template <int KTMAX, int GROUP, int ROUNDS>
__global__ SDA_N31_LB(KTMAX) void packed_gen_n31_kernel(GenLayout L, uint32_t n, uint32_t k, uint32_t t, ModParams mod,
                                                                  N31Params np, MatArg M, DrbgKey key, uint64_t chunks,
                                                                  uint64_t batches, bool vec) {
    packed_gen_n31_body<KTMAX, GROUP, ROUNDS>(L, n, k, t, mod, np, reinterpret_cast<const int32_t*>(&M.e[0]), key, chunks, batches, vec,
                                              blockIdx.x);
}

// the dual-role launch (share-gen of tile i + clerk-sum of tile i-1 in one grid) for the narrow kernel
template <int KTMAX, int GROUP, int ROUNDS>
__global__ SDA_N31_LB(KTMAX) void fused_packed_n31_kernel(GenLayout L, uint32_t n, uint32_t k, uint32_t t, ModParams mod,
                                                                    N31Params np, MatArg M, DrbgKey key, uint64_t chunks,
                                                                    uint64_t batches, FuseArgs F) {
    uint64_t idx;
    if (!fuse_dispatch(F, blockIdx.x, idx))
        packed_gen_n31_body<KTMAX, GROUP, ROUNDS>(L, n, k, t, mod, np, reinterpret_cast<const int32_t*>(&M.e[0]), key, chunks, batches,
                                                  true, idx);
}

// ---- reveal over a narrow prime: secrets[k] = R[k x n'] * sums[n'] per batch, the structure of packed_reconstruct_vec_kernel
// (share pairs in registers, LDS-staged coalesced stores) on one-limb arithmetic.  R31: centred Montgomery-form (R = 2^32)
// constants, int32 [k][n_rows] in device memory (wave-uniform: scalar loads).
template <int NMAX, int GROUP>
__global__ __launch_bounds__(kThreads) void packed_reconstruct_n31_kernel(const int64_t* __restrict__ shares, size_t row_stride,
                                                                          uint32_t n_rows, uint32_t k, size_t batches, size_t dimension,
                                                                          ModParams mod, N31Params np, const int32_t* __restrict__ R31,
                                                                          int64_t* __restrict__ out) {
    extern __shared__ int64_t stage[];                    // [2 * kThreads][k] = the block's secrets in output order
    const size_t b0 = 2 * ((size_t)blockIdx.x * kThreads + threadIdx.x);
    int32_t v0[NMAX], v1[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        uint64_t a = 0, b = 0;
        if ((uint32_t)c < n_rows && b0 < batches) {
            if (b0 + 1 < batches) {
                const ll2 v = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(shares + (size_t)c * row_stride + b0));
                a = canon_i64(v.x, mod.m, mod.mu); b = canon_i64(v.y, mod.m, mod.mu);
            } else {
                a = canon_i64(shares[(size_t)c * row_stride + b0], mod.m, mod.mu);
            }
        }
        v0[c] = n31_centre(a, np); v1[c] = n31_centre(b, np);
    }
    for (uint32_t e = 0; e < k; ++e) {
        uint32_t r0 = 0, r1 = 0;
#pragma unroll
        for (int g0 = 0; g0 < NMAX; g0 += GROUP) {
            if ((uint32_t)g0 < n_rows) {                  // wave-uniform
                int64_t S0 = 0, S1 = 0;
#pragma unroll
                for (int c = g0; c < g0 + GROUP && c < NMAX; ++c) {
                    if ((uint32_t)c < n_rows) {
                        const int32_t m = R31[(size_t)e * n_rows + c];
                        S0 += (int64_t)m * v0[c]; S1 += (int64_t)m * v1[c];
                    }
                }
                const uint32_t t0 = n31_redc(S0, np), t1 = n31_redc(S1, np);
                uint32_t s = r0 + t0, d = s - np.p;
                r0 = d < s ? d : s;
                s = r1 + t1; d = s - np.p;
                r1 = d < s ? d : s;
            }
        }
        stage[(size_t)(2 * threadIdx.x) * k + e] = (int64_t)r0;
        stage[(size_t)(2 * threadIdx.x + 1) * k + e] = (int64_t)r1;
    }
    __syncthreads();
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
