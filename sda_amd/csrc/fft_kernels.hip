// Packed-Shamir share generation in tss's own form (packed_shamir.rs:42 -> tss::packed::PackedSecretSharing::share,
// SURVEY.md App. B): values [0, secrets, draws] -> radix-2 inverse transform over the k+t+1 secret nodes ->
// zero-extension -> radix-3 forward transform over the n+1 share points; shares = evaluations 1..n.
//
// This is the path for LARGE tss-valid shapes (k + t > 32, e.g. tss's PSS_155_728_100: k=100, t=155, n=728), where the
// dense n x (k+t) matrix form costs n (k+t) multiply-accumulates per batch (185,640) against ~2,200 butterflies here.
// One workgroup owns a group of G batches (8 = the batches one CSPRNG block serves; 4, 2 or 1 of one block group when 8 do
// not leave two workgroups per CU their LDS); values live in LDS as lazily reduced UNSIGNED 64-bit numbers.
//
// Arithmetic (round 3; exactness and every bound: tests/test_fft_model.py).  p < 2^62, so 4p < 2^64: values are kept in
// [0, 4p) ("relaxed") or [0, 2p) ("reduced") and a butterfly needs a conditional subtraction of 2p only where a third
// term would pass 4p.  Every multiplication is by a table constant w with its precomputed companion w' = floor(w 2^64 / p)
// (Shoup / Harvey): q = hi64(x w'), x w - q p (low 64 bits) lies in [0, 2p) for ANY 64-bit x - one exact 64x64 high
// product and two low products, ~16 VALU instructions against ~30 for the balanced-limb Montgomery product round 2 used
// here, and its operand needs no reduction first.
//
// Structure.  The zero-extended vector has m2 = k+t+1 non-zero coefficients out of m3 = n+1, and the first two radix-3
// levels (decimation in time, digit-reversed input) only ever see one coefficient per 3-block, so they are folded into
// the scatter: one work item scales <= 9 coefficients and writes a finished 9-block (12 multiplications instead of the
// 18 + 9 of two dense levels + a scaling pass, no zero fill).  The remaining levels run two at a time in registers
// (radix 9: 9 loads, 18 multiplications, 9 stores per item), with a single radix-3 level first when their number is
// odd.  Twiddles and their companions sit in LDS (16-byte reads) when the group's values leave room (G = 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "chacha.hpp"
#include "capi_internal.hpp"
#include "drbg_lane.hpp"
#include "kernels.hpp"
#include "modarith.hpp"

namespace sda {

// Two instantiations of one kernel: WIDE (uint64_t values, p < 2^62) and NARROW (uint32_t values, p < 2^30 - the reference's
// own domain: tss's shipped primes are 20- to 23-bit).  A field value is V, a table entry (w, companion) is W.
template <typename V> struct FftW;
template <> struct FftW<uint64_t> { typedef ulonglong2 type; };
template <> struct FftW<uint32_t> { typedef uint2 type; };
template <typename V>
struct FftConst {
    V p, p2, np;             // modulus, 2p, 2^w - p (w = width of V)
    V om, oms;               // omega_shares^(m3/3) (a primitive cube root of unity) and its companion
};

// WIDE.  x (any 64-bit value) times the constant w < p, given ws = floor(w 2^64 / p) and np = 2^64 - p: congruent to x w, in
// [0, 2p).  q = hi64(x ws); the low 64 bits of x w - q p = x w + q np are ONE pair of accumulating 32 x 32 -> 64 products
// plus four low products into the high word (16 VALU instructions as compiled, 20 in the plain x * w - q * p form)
__device__ __forceinline__ uint64_t f_mulS(uint64_t x, uint64_t w, uint64_t ws, const FftConst<uint64_t>& c) {
    const uint64_t np = c.np;
    const uint64_t q = __umul64hi(x, ws);
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
    const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32), n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    const uint64_t t = (uint64_t)x0 * w0 + (uint64_t)q0 * n0;
    const uint32_t h = x0 * w1 + x1 * w0 + q0 * n1 + q1 * n0;
    return t + ((uint64_t)h << 32);
}
// x < 2m -> x < m (m = 2p: [0, 4p) -> [0, 2p); m = p: [0, 2p) -> canonical).  The borrow of the 64-bit subtraction selects:
// four instructions, where the compare-and-select form hipcc emits takes five
__device__ __forceinline__ uint64_t f_csub(uint64_t x, uint64_t m) {
    uint32_t lo, hi;
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), ml = (uint32_t)m, mh = (uint32_t)(m >> 32);
    asm("v_sub_co_u32 %0, vcc, %2, %4\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
        : "=&v"(lo), "=&v"(hi) : "v"(xl), "v"(xh), "v"(ml), "v"(mh) : "vcc");
    return ((uint64_t)hi << 32) | lo;
}
// NARROW (p < 2^30, so 4p < 2^32: the same lazy ranges in 32 bits).  Shoup product with ws = floor(w 2^32 / p): q = hi32(x ws),
// x w - q p in the low 32 bits lies in [0, 2p) for ANY 32-bit x - three multiplications and a subtraction (WIDE: 16
// instructions); the conditional subtraction is a subtraction and an unsigned minimum (x - m wraps exactly when x < m).
__device__ __forceinline__ uint32_t f_mulS(uint32_t x, uint32_t w, uint32_t ws, const FftConst<uint32_t>& c) {
    const uint32_t q = __umulhi(x, ws);
    return x * w - q * c.p;
}
__device__ __forceinline__ uint32_t f_csub(uint32_t x, uint32_t m) {
    const uint32_t d = x - m;
    return d < x ? d : x;
}
template <typename V> __device__ __forceinline__ V f_red2(V x, V p2) { return f_csub(x, p2); }

// radix-3 butterfly on A, B, C (B, C in [0, 2p): already multiplied by their twiddles): y_d = A + w^d B + w^2d C.  A in [0, 2p)
// gives outputs in [0, 4p).  LAZY (narrow values and (4 b + 4) p < 2^32, b = the number of radix-3 levels: tss's 20- to 23-bit
// primes): no conditional subtraction at all - every level lets the A chain grow by 4p (B and C always come out of a Shoup
// product, which takes ANY 32-bit operand and returns [0, 2p)), and the last pass reduces once (tests/test_narrow_model.py)
template <bool LAZY, typename V>
__device__ __forceinline__ void f_r3(V A, V Bv, V Cv, const FftConst<V>& c, V& y0, V& y1, V& y2) {
    const V w = f_mulS((V)(Bv + c.p2 - Cv), c.om, c.oms, c);          // w (B - C); w^2 = -1 - w
    if constexpr (LAZY) {
        y0 = A + Bv + Cv;
        y1 = A + (c.p2 - Cv) + w;
        y2 = A + (c.p2 - Bv) + (c.p2 - w);
    } else {
        y0 = f_red2((V)(A + Bv), c.p2) + Cv;
        y1 = f_red2((V)(A + c.p2 - Cv), c.p2) + w;                    // A - C + w (B - C)
        y2 = f_red2((V)(A + c.p2 - Bv), c.p2) + (c.p2 - w);           // A - B - w (B - C)
    }
}
// the A input of a level: [0, 4p) -> [0, 2p), or left to grow (LAZY)
template <bool LAZY, typename V> __device__ __forceinline__ V f_redA(V x, const FftConst<V>& c) {
    if constexpr (LAZY) return x; else return f_red2(x, c.p2);
}
// any value of the lazy chain -> [0, 2p): a Shoup product by 1 (ones = floor(2^32 / p))
template <bool LAZY, typename V> __device__ __forceinline__ V f_full2(V x, V ones, const FftConst<V>& c) {
    if constexpr (LAZY) return f_mulS(x, (V)1, ones, c); else return f_red2(x, c.p2);
}

__device__ __forceinline__ uint32_t f_bitrev(uint32_t i, uint32_t bits) { return bits ? __brev(i) >> (32 - bits) : 0u; }
__device__ __forceinline__ uint32_t f_trirev(uint32_t i, uint32_t digits) {
    uint32_t r = 0;
    for (uint32_t d = 0; d < digits; ++d) {
        const uint32_t q = __umulhi(i, 0x55555556u);                 // i / 3, exact for i < 2^31
        r = r * 3u + (i - 3u * q);
        i = q;
    }
    return r;
}
// exact floor(x / d) for x < 2^16, magic = floor(2^32 / d) + 1 (d > 1), identity for d = 1
__device__ __forceinline__ uint32_t f_div(uint32_t x, uint32_t d, uint32_t magic) { return d > 1 ? __umulhi(x, magic) : x; }

// TWL: the twiddle tables (with their companions) are copied to LDS; otherwise they are read from global memory
// __launch_bounds__(1024): the launcher uses 512 threads (1024 for the one-batch shapes, any multiple of 64 through the A/B
// knob), so the register budget is the 1024-thread one (128 VGPRs) for every instance - all of them compile to 94 - 106
// without spills, and occupancy is set by the LDS (two workgroups per CU), not by registers.
template <int ROUNDS, bool TWL, typename V, bool LAZY>
__global__ __launch_bounds__(1024) void packed_gen_fft_kernel(GenLayout L, ModParams mod, DrbgKey key, FftPlan F, uint64_t groups, uint64_t batches) {
    typedef typename FftW<V>::type W;
    extern __shared__ __align__(16) uint64_t lds_raw[];
    V* lds = reinterpret_cast<V*>(lds_raw);
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    const uint32_t G = F.G, m2 = F.m2, m3 = F.m3, k = F.k, t = F.t;
    // Fewer than 8 batches per workgroup: a 128-byte line of a clerk row holds the 8-byte shares of 16 consecutive batches,
    // i.e. pieces from 16 / G DIFFERENT workgroups, which can only be merged in an L2 - and workgroup b runs on XCD b mod 8,
    // each XCD with an L2 of its own.  The launcher pads the groups of a participant to a multiple of 8 * (16 / G) and the
    // group index is permuted so that the 16 / G workgroups of one line sit on ONE XCD, dispatched within 128 block indices
    // of each other (tss's PSS_155_19682_100: 0.17 -> see DESIGN.md; the pieces used to leave eight different L2s as partial
    // lines, 16 fabric writes per line).
    const uint64_t gpad = F.groups_padded ? F.groups_padded : groups;
    const uint64_t p = blockIdx.x / gpad;
    uint64_t g = blockIdx.x - p * gpad;
    if (F.groups_padded) {
        const uint32_t C = 16u / G;
        const uint64_t xcd = g & 7u, slot = g >> 3, chunk = slot / C, i = slot - chunk * C;
        g = (chunk * 8u + xcd) * C + i;
        if (g >= groups) return;
    }
    // LDS: [twiddles of the radix-3 part | twiddles of the radix-2 part |] X [G][m2] | Y [G][m3]
    const W* tw3 = TWL ? reinterpret_cast<const W*>(lds) : reinterpret_cast<const W*>(F.tw3);
    const W* tw2 = TWL ? reinterpret_cast<const W*>(lds) + m3 : reinterpret_cast<const W*>(F.tw2);
    V* X = lds + (TWL ? 2 * ((size_t)m3 + (m2 >> 1)) : 0);          // secret-node values / coefficients
    V* Y = X + (size_t)G * m2;                                      // share-point values
    FftConst<V> c;
    c.p = (V)mod.m; c.p2 = (V)(2 * mod.m); c.np = (V)((V)0 - (V)mod.m); c.om = (V)F.omega; c.oms = (V)F.omega_s;
    const V scale = (V)F.scale, scale_s = (V)F.scale_s, ones = (V)F.one_s;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    const uint64_t stream = L.first_participant + p;
    const uint64_t b_first = g * G;

    if (TWL) {
        W* dst = reinterpret_cast<W*>(lds);
        const W* s3 = reinterpret_cast<const W*>(F.tw3);
        const W* s2 = reinterpret_cast<const W*>(F.tw2);
        for (uint32_t u = tid; u < m3; u += T) dst[u] = s3[u];
        for (uint32_t u = tid; u < (m2 >> 1); u += T) dst[m3 + u] = s2[u];
    }

    // ---- values: [0, secrets (zero-padded, batched.rs:37-43), draws], canonical -----------------------------------
    for (uint32_t u = tid; u < G * (k + 1); u += T) {
        const uint32_t j = f_div(u, k + 1, F.magic_k1), i = u - j * (k + 1);
        const uint64_t b = b_first + j;
        uint64_t v = 0;
        if (i > 0) {
            const uint64_t e = b * k + (i - 1);
            if (b < batches && e < L.len) v = canon_i64(sp[e], mod.m, mod.mu);
        }
        X[(size_t)j * m2 + i] = (V)v;
    }
    if (rp) {
        for (uint32_t u = tid; u < G * t; u += T) {
            const uint32_t j = f_div(u, t, F.magic_t), i = u - j * t;
            const uint64_t b = b_first + j;
            X[(size_t)j * m2 + 1 + k + i] = (V)(b < batches ? canon_i64(rp[b * t + i], mod.m, mod.mu) : 0);
        }
    } else if (drbg_paired(mod.m)) {          // the paired rule (modarith.hpp): one block = draws 2j, 2j + 1 of 8 consecutive batches
        const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
        const uint32_t t2 = (t + 1u) >> 1;
        const uint32_t per = G >= 8 ? 8u : G, off = G >= 8 ? 0u : (uint32_t)(b_first & 7);       // batches of a block group this workgroup holds
        const uint32_t blocks = G >= 8 ? (G >> 3) : 1u;
        for (uint32_t u = tid; u < blocks * t2; u += T) {
            const uint32_t nb = u / t2, j = u - nb * t2;
            const uint64_t grp = G >= 8 ? (g * (G >> 3) + nb) : (b_first >> 3);
            const uint64_t I = grp * (uint64_t)t2 + j;
            uint32_t o[16];
            chacha_block_lane<ROUNDS>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
            for (uint32_t jj = 0; jj < per; ++jj) {
                const uint32_t w8 = off + jj, want_hi = 8u * (w8 & 1u) + (w8 >> 1), want_lo = want_hi + 4u;
                uint32_t hi = 0, lo = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) {                              // o[] lives in registers: select, do not index
                    if ((uint32_t)w == want_hi) hi = o[w];
                    if ((uint32_t)w == want_lo) lo = o[w];
                }
                const uint64_t bl = 8u * nb + jj;                          // batch inside the workgroup's group
                const uint64_t pr = f_draw_pair<ROUNDS>(((uint64_t)hi << 32) | lo, kk, stream, (b_first + bl) * (uint64_t)t2 + j, mod.m, mod.lemire_thr2);
                X[(size_t)bl * m2 + 1 + k + 2u * j] = (V)(uint32_t)pr;
                if (2u * j + 1u < t) X[(size_t)bl * m2 + 2 + k + 2u * j] = (V)(uint32_t)(pr >> 32);
            }
        }
    } else if (G >= 8) {                      // one CSPRNG block serves draw i of 8 consecutive batches; G / 8 such blocks of 8
        const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
        for (uint32_t u = tid; u < (G >> 3) * t; u += T) {
            const uint32_t nb = f_div(u, t, F.magic_t), i = u - nb * t;
            const uint64_t I = (g * (G >> 3) + nb) * (uint64_t)t + i;
            uint32_t o[16];
            chacha_block_lane<ROUNDS>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int cc = jj >> 1, e = jj & 1;
                const uint64_t xw = ((uint64_t)o[8 * e + cc] << 32) | o[8 * e + 4 + cc];
                uint64_t val;
                if (!f_lemire<V>(xw, mod, val))
                    val = f_drbg_retry<ROUNDS>(kk[0], kk[1], kk[2], kk[3], kk[4], kk[5], kk[6], kk[7], stream, (b_first + 8u * nb + jj) * (uint64_t)t + i, mod.m, mod.lemire_thr);
                X[(size_t)(8u * nb + jj) * m2 + 1 + k + i] = (V)val;
            }
        }
    } else {                                  // G = 1, 2, 4 batches of ONE block group of 8: each block serves G of its 8 word pairs
        const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
        const uint32_t off = (uint32_t)(b_first & 7);                       // a multiple of G
        for (uint32_t i = tid; i < t; i += T) {
            const uint64_t I = (b_first >> 3) * (uint64_t)t + i;
            uint32_t o[16];
            chacha_block_lane<ROUNDS>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
            for (uint32_t jj = 0; jj < G; ++jj) {
                const uint32_t w8 = off + jj, want_hi = 8u * (w8 & 1u) + (w8 >> 1), want_lo = want_hi + 4u;
                uint32_t hi = 0, lo = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) {                              // o[] lives in registers: select, do not index
                    if ((uint32_t)w == want_hi) hi = o[w];
                    if ((uint32_t)w == want_lo) lo = o[w];
                }
                uint64_t val;
                if (!f_lemire<V>(((uint64_t)hi << 32) | lo, mod, val))
                    val = f_drbg_retry<ROUNDS>(kk[0], kk[1], kk[2], kk[3], kk[4], kk[5], kk[6], kk[7], stream, (b_first + jj) * (uint64_t)t + i, mod.m, mod.lemire_thr);
                X[(size_t)jj * m2 + 1 + k + i] = (V)val;
            }
        }
    }
    __syncthreads();

    // ---- radix-2 inverse transform, decimation in frequency: natural order in, bit-reversed order out; values in [0, 2p).
    // A single radix-2 level first when the number of levels is odd, then two levels at a time in registers (radix 4).
    uint32_t mblk = m2, lg = F.a;
    if (lg & 1u) {
        const uint32_t h = mblk >> 1;
        for (uint32_t u = tid; u < G * h; u += T) {
            const uint32_t j = u >> (lg - 1), jj = u & (h - 1);
            V* x = X + (size_t)j * m2 + jj;
            const V av = x[0], bv = x[h];
            const W w = tw2[jj];
            x[0] = f_red2(av + bv, c.p2);
            x[h] = f_mulS(av + c.p2 - bv, w.x, w.y, c);                  // the table's entry 0 is 1 with its companion
        }
        __syncthreads();
        mblk >>= 1; --lg;
    }
    for (; lg >= 2; lg -= 2, mblk >>= 2) {
        const uint32_t qd = mblk >> 2, step = m2 / mblk;                   // quarter of a block; twiddle stride of the outer level
        const uint32_t items = m2 >> 2;                                     // per batch
        for (uint32_t u = tid; u < G * items; u += T) {
            const uint32_t j = u >> (F.a - 2), i = u & (items - 1);
            const uint32_t blk = i >> (lg - 2), jj = i & (qd - 1);
            V* x = X + (size_t)j * m2 + (size_t)blk * mblk + jj;
            const V x0 = x[0], x1 = x[qd], x2 = x[2 * qd], x3 = x[3 * qd];
            // level with blocks of mblk: (x0, x2) and (x1, x3), twiddles w^jj and w^(jj + qd)
            const V a0 = f_red2(x0 + x2, c.p2), a1 = f_red2(x1 + x3, c.p2);
            const W wb = tw2[(jj + qd) * step];
            const V a3 = f_mulS(x1 + c.p2 - x3, wb.x, wb.y, c);
            V a2, b1, b3;
            if (qd > 1) {
                const W wa = tw2[jj * step], wc = tw2[2 * jj * step];
                a2 = f_mulS(x0 + c.p2 - x2, wa.x, wa.y, c);
                // level with blocks of mblk / 2: (a0, a1) and (a2, a3), twiddle (w^2)^jj
                b1 = f_mulS(a0 + c.p2 - a1, wc.x, wc.y, c);
                b3 = f_mulS(a2 + c.p2 - a3, wc.x, wc.y, c);
            } else {                                                        // the last pass: jj = 0, those twiddles are 1
                a2 = f_red2(x0 + c.p2 - x2, c.p2);
                b1 = f_red2(a0 + c.p2 - a1, c.p2);
                b3 = f_red2(a2 + c.p2 - a3, c.p2);
            }
            x[0] = f_red2(a0 + a1, c.p2);
            x[qd] = b1;
            x[2 * qd] = f_red2(a2 + a3, c.p2);
            x[3 * qd] = b3;
        }
        __syncthreads();
    }

    // ---- scale by 1 / m2, zero-extend, and the first TWO radix-3 levels (decimation in time, digit-reversed input) -----
    // 9-block q of Y takes the coefficients r + e1 S2 + e0 S1 (r = the digit reversal of q over b - 2 digits, S1 = m3 / 3,
    // S2 = m3 / 9), at position e0 + 3 e1 of the block; those at or beyond m2 are the zero extension.
    const uint32_t ninth = m3 / 9, S1 = m3 / 3, S2 = ninth;
    const uint32_t magic_ninth = ninth > 1 ? (uint32_t)(0x100000000ull / ninth) + 1u : 0u;
    const uint32_t nz = F.nz_mask;
    for (uint32_t u = tid; u < G * ninth; u += T) {
        const uint32_t j = f_div(u, ninth, magic_ninth), q = u - j * ninth;
        const uint32_t r = f_trirev(q, F.b - 2);
        const V* xj = X + (size_t)j * m2;
        V v[3][3];                                                  // [e1][d]: level-1 outputs
#pragma unroll
        for (int e1 = 0; e1 < 3; ++e1) {
            V in[3];
#pragma unroll
            for (int e0 = 0; e0 < 3; ++e0) {
                in[e0] = 0;
                if (nz >> (3 * e0 + e1) & 1u) {                            // uniform: some block has this coefficient
                    const uint32_t ci = r + (uint32_t)e1 * S2 + (uint32_t)e0 * S1;
                    if (ci < m2) in[e0] = f_mulS(xj[f_bitrev(ci, F.a)], scale, scale_s, c);
                }
            }
            if ((nz >> (3 + e1) & 1u) || (nz >> (6 + e1) & 1u)) {          // uniform
                f_r3<LAZY>(in[0], in[1], in[2], c, v[e1][0], v[e1][1], v[e1][2]);
#pragma unroll
                for (int d = 0; d < 3; ++d) v[e1][d] = f_redA<LAZY>(v[e1][d], c);
            } else {
                v[e1][0] = v[e1][1] = v[e1][2] = in[0];                    // two of three inputs are zero-extension zeros
            }
        }
        V* y = Y + (size_t)j * m3 + 9u * q;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            V Bv = v[1][jj], Cv = v[2][jj];
            if (jj) {
                const W w1 = tw3[jj * ninth], w2 = tw3[2 * jj * ninth];
                Bv = f_mulS(Bv, w1.x, w1.y, c);
                Cv = f_mulS(Cv, w2.x, w2.y, c);
            } else if (LAZY) {                                                // twiddle 1: no product brings them back to [0, 2p)
                Bv = f_full2<LAZY>(Bv, ones, c);
                Cv = f_full2<LAZY>(Cv, ones, c);
            }
            f_r3<LAZY>(v[0][jj], Bv, Cv, c, y[jj], y[jj + 3], y[jj + 6]);
        }
    }
    __syncthreads();

    // ---- remaining radix-3 levels: a single one when their number is odd, then two at a time ---------------------------
    uint32_t t3 = 9, left = F.b - 2;
    if (left & 1u) {
        const uint32_t step = m3 / (3 * t3);
        const uint32_t magic_third = (uint32_t)(0x100000000ull / S1) + 1u, magic = (uint32_t)(0x100000000ull / t3) + 1u;
        for (uint32_t u = tid; u < G * S1; u += T) {
            const uint32_t j = f_div(u, S1, magic_third), q = u - j * S1;
            const uint32_t blk = f_div(q, t3, magic), jj = q - blk * t3;
            V* y = Y + (size_t)j * m3 + (size_t)blk * (3 * t3) + jj;
            const W w1 = tw3[jj * step], w2 = tw3[2 * jj * step];
            const V A = f_redA<LAZY>(y[0], c);
            const V Bv = f_mulS(y[t3], w1.x, w1.y, c), Cv = f_mulS(y[2 * t3], w2.x, w2.y, c);
            f_r3<LAZY>(A, Bv, Cv, c, y[0], y[t3], y[2 * t3]);
        }
        __syncthreads();
        t3 *= 3; --left;
    }
    int64_t* op = L.out + p * L.out_stride_participant;
    bool stored = false;
    for (; left; left -= 2, t3 *= 9) {
        const uint32_t step_a = m3 / (3 * t3), step_b = m3 / (9 * t3);
        const uint32_t magic = (uint32_t)(0x100000000ull / t3) + 1u;
        // the LAST pass (one block of 9 t3 = m3 per batch) hands its evaluations straight to global memory: items are then
        // dealt batch-fastest, so that G neighbouring lanes store the G batches of one clerk row (64 contiguous bytes), with
        // non-temporal stores: write-back stores cost 1.35x the share bytes in WRITE_SIZE (lines written back more than
        // once), these 1.02x - measured, also with whole 128-byte lines per row (16 batches per workgroup), same figures
        const bool last = left == 2;
        for (uint32_t u = tid; u < G * ninth; u += T) {
            uint32_t j, q;
            if (last && G > 1) { j = u & (G - 1); q = u >> F.lgG; }
            else { j = f_div(u, ninth, magic_ninth); q = u - j * ninth; }
            const uint32_t blk = f_div(q, t3, magic), jj = q - blk * t3;
            V* y = Y + (size_t)j * m3 + (size_t)blk * (9 * t3) + jj;
            V a[9], v[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) a[e] = y[(uint32_t)e * t3];
            {   // level with blocks of 3 t3: the three butterflies share their twiddles
                const W w1 = tw3[jj * step_a], w2 = tw3[2 * jj * step_a];
#pragma unroll
                for (int e1 = 0; e1 < 3; ++e1) {
                    const V A = f_redA<LAZY>(a[3 * e1], c);
                    const V Bv = f_mulS(a[3 * e1 + 1], w1.x, w1.y, c), Cv = f_mulS(a[3 * e1 + 2], w2.x, w2.y, c);
                    f_r3<LAZY>(A, Bv, Cv, c, v[3 * e1], v[3 * e1 + 1], v[3 * e1 + 2]);
                }
            }
            V o[9];
#pragma unroll
            for (int d = 0; d < 3; ++d) {   // level with blocks of 9 t3: element jj + d t3 of each third
                const uint32_t jb = jj + (uint32_t)d * t3;
                const W w1 = tw3[jb * step_b], w2 = tw3[2 * jb * step_b];
                const V A = f_redA<LAZY>(v[d], c);
                const V Bv = f_mulS(v[3 + d], w1.x, w1.y, c), Cv = f_mulS(v[6 + d], w2.x, w2.y, c);
                f_r3<LAZY>(A, Bv, Cv, c, o[d], o[d + 3], o[d + 6]);
            }
            if (!last) {
#pragma unroll
                for (int e = 0; e < 9; ++e) y[(uint32_t)e * t3] = o[e];
            } else {
                // shares = evaluations 1..n, canonical, clerk-major (batched.rs:46-48); evaluation 0 is f(1) = 0
                const uint64_t b = b_first + j;
                if (b < batches) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) {
                        const uint32_t pos = jj + (uint32_t)e * t3;
                        if (pos) {
                            const int64_t val = (int64_t)(uint64_t)f_csub(f_full2<LAZY>(o[e], ones, c), c.p);
                            int64_t* dst = op + (size_t)(pos - 1) * L.out_stride_clerk + b;
                            // 64-byte segments (8 batches per row and workgroup): non-temporal.  Fewer batches per workgroup
                            // leave 8- to 32-byte pieces of a line to DIFFERENT workgroups: those must meet in the L2
                            // (write-back) - as non-temporal partial writes they cost 30000x (PSS_155_19682_100, measured)
                            if (G >= 8) __builtin_nontemporal_store(val, dst); else *dst = val;
                        }
                    }
                }
            }
        }
        if (!last) __syncthreads();
        stored = last;
    }
    if (stored) return;

    // ---- shares = evaluations 1..n from LDS (shapes without a radix-9 pass), canonical, clerk-major ------------------------
    for (uint32_t u = tid; u < G * F.n; u += T) {
        const uint32_t sj = u >> F.lgG, jj = u & (G - 1);   // batch fastest: G consecutive values per clerk row
        const uint64_t b = b_first + jj;
        if (b >= batches) continue;
        const int64_t val = (int64_t)(uint64_t)f_csub(f_full2<LAZY>(Y[(size_t)jj * m3 + sj + 1], ones, c), c.p);
        int64_t* dst = op + (size_t)sj * L.out_stride_clerk + b;
        if (G >= 8) __builtin_nontemporal_store(val, dst); else *dst = val;
    }
}

size_t fft_lds_bytes(uint32_t m2, uint32_t m3, uint32_t G, bool tw_lds, bool narrow) {
    return ((size_t)G * ((size_t)m2 + m3) + (tw_lds ? 2 * ((size_t)m3 + m2 / 2) : 0)) * (narrow ? 4 : 8);
}

template <typename V, bool LAZY>
static hipError_t fft_launch_v(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const FftPlan& F, int rounds, hipStream_t s) {
    const uint64_t batches = (L.len + F.k - 1) / F.k;
    const uint64_t groups = (batches + F.G - 1) / F.G;
    if (groups * L.participants == 0) return hipSuccess;
    const size_t lds = fft_lds_bytes(F.m2, F.m3, F.G, F.tw_lds != 0, sizeof(V) == 4);
    // 512 threads = 4 waves per SIMD with the two workgroups a CU's LDS holds (PSS_155_728_100, round 2: 58 ms per 500 x 1 Mi
    // tile against 66 with 256 threads, 79 with 1024, 120 with 128)
    const long want_threads = F.want_threads;                                   // A/B only (knob SDA_FFT_THREADS when the handle was created)
    unsigned threads = (F.G == 1 && F.m3 > 2187) ? 1024u : 512u;
    if (want_threads >= 64 && want_threads <= 1024 && want_threads % 64 == 0) threads = (unsigned)want_threads;
    if (rounds != 20 && rounds != 12 && rounds != 8) return hipErrorInvalidValue;
    auto kern = F.tw_lds ? (rounds == 20 ? packed_gen_fft_kernel<20, true, V, LAZY> : rounds == 12 ? packed_gen_fft_kernel<12, true, V, LAZY> : packed_gen_fft_kernel<8, true, V, LAZY>)
                         : (rounds == 20 ? packed_gen_fft_kernel<20, false, V, LAZY> : rounds == 12 ? packed_gen_fft_kernel<12, false, V, LAZY> : packed_gen_fft_kernel<8, false, V, LAZY>);
    if (lds > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    FftPlan Fl = F;
    Fl.groups_padded = 0;
    uint64_t gridg = groups;
    if (F.G < 8 && !F.no_xcd_map) {
        const uint64_t unit = 8ull * (16u / F.G);
        gridg = (groups + unit - 1) / unit * unit;
        Fl.groups_padded = gridg;
    }
    const uint64_t max_blocks = 0x7FFFFFFFull;
    uint64_t per = max_blocks / gridg;
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
        note_kernel("packed_gen_fft_kernel<%d, %s, %s, %s>", rounds, F.tw_lds ? "true" : "false", sizeof(V) == 4 ? "unsigned int" : "unsigned long",
                    LAZY ? "true" : "false");
        kern<<<dim3((unsigned)(gridg * cnt)), dim3(threads), lds, s>>>(S, mod, key, Fl, groups, batches);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_packed_generate_fft(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const FftPlan& F, int rounds,
                                      hipStream_t s) {
    if (!F.narrow) return fft_launch_v<uint64_t, false>(L, mod, key, F, rounds, s);
    return F.lazy ? fft_launch_v<uint32_t, true>(L, mod, key, F, rounds, s) : fft_launch_v<uint32_t, false>(L, mod, key, F, rounds, s);
}

}  // namespace sda
