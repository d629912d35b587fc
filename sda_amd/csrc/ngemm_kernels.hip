// Packed-Shamir share generation over a NARROW prime (p <= 0x7F7F7F, just below 2^23: tss's shipped 746497 and 5038849,
// full_loop.rs's 433) for LARGE shapes (k + t > 16, e.g. tss's PSS_155_728_100) as a limb GEMM on the matrix cores.
//
// packed_shamir.rs:42 -> tss share(): shares = M [secrets ; draws], M the n x (k + t) matrix of the polynomial through
// (1, 0), the secrets at omega_secrets^(1..k) and the draws at omega_secrets^(k+1..k+t), evaluated at omega_shares^(1..n).
// The transform kernel (fft_kernels.hip) computes this product with tss's own radix-2 / radix-3 structure in ~15 vector
// instructions per secret and is bound by the vector ALUs at 0.42 of the HBM roofline.  Over a prime this small the DENSE
// product is cheaper on this chip: a residue below 2^23 is THREE balanced base-256 digits (int8; the matrix entries are
// centred first, the values are taken as they are), so
//     column c = sum over terms and la + lb = c of digit_la(M) digit_lb(value)          (c = 0..4)
// is nine v_mfma_i32_16x16x64_i8 per 16 shares x 16 batches x 64 terms - 2.4 M multiply-adds per batch of PSS_155_728_100 at
// 16384 per instruction - and the five 32-bit column sums of one share come back to ONE residue with five v_mad_i64_i32 by
// c_j = 256^j 2^32 mod p and one three-instruction Montgomery reduction (R = 2^32).  Per secret: ~17 matrix-core cycles per
// SIMD and ~1.5 vector instructions of epilogue, beside the t / k ChaCha20 draws that every form of this path pays.
//
// Work decomposition.  ONE workgroup per CU: 8 compute waves (two per SIMD) + 1 loader wave.  The workgroup owns 128 NT
// consecutive batches of one participant; compute wave w holds the value digits of its 16 NT batches for ALL terms in
// registers (B operands: NT x KS x 3 fragments) and the workgroup sweeps the row tiles of the matrix, whose digits the host
// laid out fragment by fragment ([row tile][64-term step][digit][lane] 16 bytes).  The LOADER wave streams those tiles
// global -> LDS with global_load_lds_dwordx4 into a ring of slots, two tiles ahead, with counted s_waitcnt vmcnt - it has no
// stores in its in-order counter, so no tile ever waits for share stores to be acknowledged; the compute waves read the
// fragments back with conflict-free 16-byte LDS loads.  Column c of batch tile nt is batch NT c + nt of the wave, so a lane owns
// adjacent batch columns and shares leave as 16-byte non-temporal buffer stores (row pointer in a scalar descriptor, one 32-bit
// lane offset), 256-byte row segments.  The B fragments of the last two steps are read from LDS during the row tiles (the last
// step's digit tile and a per-wave stash, NgBSource): the row loop must fit the 168 registers three waves per SIMD leave WITHOUT
// scratch memory - a reload there waits for every share store in flight (tests/test_ngemm_isa.py reads the assembly).
// Round 6: THREE MORE WAVES per workgroup, the clerk waves, sum the PREVIOUS tile's shares while all this runs (ng_clerk_wave).
// Values reach the B-operand layout through an LDS tile [digit][batch][64 terms], one 64-term step at a time; the draws are
// sda-drbg-v1's PAIRED rule (every prime of this kernel is in its domain): one lane = one ChaCha20 block = draws 2j and 2j + 1
// of 8 consecutive batches, the same words the transform kernel reads, so both produce identical shares from
// identical inputs and keys.  With the library's own randomness the draws ARE
// shares 0..t-1 (systematic share map, include/sda_hip.h) and the matrix has only the other n - t rows.
// Measurements behind these choices (MFMA / vector co-issue, the vmcnt trap, the half-period offset of the wave pairs):
// DESIGN.md 4 "Narrow limb GEMM".
//
// Exactness and every bound (digits, columns, the reduction's operand): tests/test_ngemm_model.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_internal.hpp"
#include "chacha.hpp"
#include "clerk_sum.hpp"
#include "drbg_lane.hpp"
#include "kernels.hpp"
#include "modarith.hpp"

namespace sda {

typedef int ng_v4i __attribute__((ext_vector_type(4)));
typedef unsigned int ng_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int ng_v2u __attribute__((ext_vector_type(2)));

static constexpr int kNgCompute = 8;                // compute waves per workgroup (two per SIMD), one workgroup per CU
static constexpr int kNgWorkers = 64 * kNgCompute;
#ifndef NG_CLERK_WAVES
#define NG_CLERK_WAVES 3                            // (tools/build_kernel_variant.sh builds A/B variants with other counts; 0 = rounds 4 - 5's nine waves)
#endif
static constexpr int kNgClerkWaves = NG_CLERK_WAVES; // clerk waves of a share-generation workgroup (round 6): one on each SIMD that does not hold the loader
static constexpr int kNgThreads = kNgWorkers + 64 + 64 * kNgClerkWaves;  // + the loader wave + the clerk waves: three waves on every SIMD
typedef __attribute__((address_space(3))) uint8_t* ng_lptr;     // LDS pointers inside the non-inlined passes: a generic pointer argument costs
                                                                 // a 64-bit address computation per access
static constexpr int kNgRow = 80;                 // bytes of one (digit, batch) row of the value tile: 64 terms + 16 (bank spread)

// c x + acc as ONE v_mad_i64_i32.  Written in plain C on purpose: the operand x comes straight out of an MFMA accumulator, and
// only instructions the compiler selects itself get the wait states that a matrix-core result needs before a vector
// instruction may read it (an inline-asm v_mad_i64_i32 placed right behind the v_mfma read stale registers: rows 4 g + 0 of
// the first batch tile, measured).  The constants sit in VGPRs (pinned once per kernel) because hipcc selects the 64-bit
// multiply-add only for vector x vector operands.
__device__ __forceinline__ int32_t ng_pin_vgpr(int32_t c) {
    asm volatile("" : "+v"(c));
    return c;
}
__device__ __forceinline__ int64_t ng_mad(int32_t v_c, int32_t x, int64_t acc) { return (int64_t)v_c * (int64_t)x + acc; }
// canonical residue v < 2^23 - 2^15 -> its three balanced base-256 digits in bytes 0..2 (two's complement): the bytes of
// (v + 0x808080) ^ 0x808080.  Values are NOT centred (only the matrix is): the column and reduction bounds hold for any digits
// in [-128, 127] (tests/test_ngemm_model.py), and a digit split of two instructions is what the staging passes can afford
__device__ __forceinline__ uint32_t ng_digits(uint32_t v) { return (v + 0x00808080u) ^ 0x00808080u; }
__device__ __forceinline__ void ng_put(ng_lptr tile, uint32_t wgb, uint32_t batch, uint32_t term, uint32_t d) {
    ng_lptr q = tile + batch * kNgRow + term;
    q[0] = (uint8_t)d;
    q[wgb * kNgRow] = (uint8_t)(d >> 8);
    q[2 * wgb * kNgRow] = (uint8_t)(d >> 16);
}
// S = sum_j C_j c_j (|S| < p 2^31) -> S 2^-32 mod p, canonical.  q = lo(S) p^-1 (signed 32 bits): q p has the low word of S, so
// (S - q p) / 2^32 = hi(S) - mulhi(q, p) with no borrow to look after - three instructions - and lies in (-p, p); the unsigned
// minimum of t and t + p is the canonical one (a negative t wraps to a huge value)
__device__ __forceinline__ uint32_t ng_redc(int64_t S, const N31Params& P) {
    const int32_t q = (int32_t)((uint32_t)S * P.pinv);              // P.pinv = +p^-1 mod 2^32 in this kernel's plan
    const uint32_t t = (uint32_t)((int32_t)(S >> 32) - __mulhi(q, (int32_t)P.p));
    const uint32_t u = t + P.p;
    return u < t ? u : t;
}

// any i64 -> canonical residue: the common case (already canonical) inline, the rest behind a call - the staging code is
// executed a few times per workgroup and has to stay small (see ng_stage)
__device__ __noinline__ uint64_t ng_canon_slow(int64_t x, uint64_t m, uint64_t mu) { return canon_i64(x, m, mu); }
__device__ __forceinline__ uint32_t ng_canon(int64_t x, const ModParams& mod) {
    return (uint64_t)x < mod.m ? (uint32_t)x : (uint32_t)ng_canon_slow(x, mod.m, mod.mu);
}

// ---- the three passes of a 64-term step of the values [secrets (zero-padded, batched.rs:37-43) ; draws ; zeros] --------------
// Each pass is ONE function (not inlined) shared by all steps: inlined per step the kernel was 121 KB of code for four steps,
// streamed once per workgroup through an instruction cache of 64 KB that two CUs share.  (A run-time loop over the steps keeps
// one copy too, but then the B fragments - indexed by the step - leave the registers: 11.8 -> 29.5 ms per tile, measured.)

// secrets / injected draws / zero padding of the step -> digit tile.  lane = term, one batch per wave and round; ALL the loads
// are issued before the first value is used (clamped addresses, no branches in between): one memory latency per step
template <int WGB>
__device__ __noinline__ void ng_load_pass(uint8_t* Bt, const int64_t* sp, const int64_t* rp, uint64_t len, uint64_t batches, uint64_t b0,
                                          uint32_t k, uint32_t t, uint32_t t_lo, uint64_t m, uint64_t mu) {
    constexpr int ROUNDS_ = WGB / kNgCompute;
    typedef const __attribute__((address_space(1))) int64_t* gptr;           // (a generic pointer argument would be read with flat loads)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, kt = k + t;
    const ModParams mod{m, mu, 0, 0};
    const uint32_t term = t_lo + lane;
    const bool is_secret = term < k, is_draw = !is_secret && term < kt;
    const bool reads = is_secret || (is_draw && rp != nullptr);
    if (is_draw && rp == nullptr) return;                                    // CSPRNG draws are written by ng_draw_pass
    ng_lptr dst = (ng_lptr)Bt + wave * kNgRow + lane;
    constexpr uint32_t PLANE = WGB * kNgRow, ROUND = kNgCompute * kNgRow;
    if (!reads) {                                                            // zero padding beyond k + t
#pragma unroll
        for (int it = 0; it < ROUNDS_; ++it) { dst[it * ROUND] = 0; dst[PLANE + it * ROUND] = 0; dst[2 * PLANE + it * ROUND] = 0; }
        return;
    }
    gptr src = (gptr)(is_secret ? sp : rp);
    const uint32_t per = is_secret ? k : t, off = is_secret ? term : term - k;
    const uint64_t lim = is_secret ? len : batches * (uint64_t)t;
    int64_t raw[ROUNDS_];
    // every batch of the workgroup and every secret of those batches exists (all workgroups but a participant's last): no
    // clamps, no per-element bounds, one pointer increment per round
    const bool full = b0 + WGB <= batches && (b0 + WGB) * (uint64_t)k <= len;
    uint64_t any = 0;
    if (full) {
        gptr q = src + (b0 + wave) * per + off;
        const size_t step = (size_t)kNgCompute * per;
#pragma unroll
        for (int it = 0; it < ROUNDS_; ++it) raw[it] = q[(size_t)it * step];
    } else {
#pragma unroll
        for (int it = 0; it < ROUNDS_; ++it) {
            const uint64_t b = b0 + (uint32_t)it * (uint32_t)kNgCompute + wave, e = b * per + off;
            raw[it] = b < batches && e < lim ? src[e] : 0;                   // zero padding (batched.rs:37-43)
        }
    }
    bool bad = false;                                                        // some value of this lane is not canonical yet
#pragma unroll
    for (int it = 0; it < ROUNDS_; ++it) {
        any |= (uint64_t)raw[it] >> 32;
        bad |= (uint32_t)raw[it] >= (uint32_t)m;                             // m < 2^23
    }
    if (bad || any != 0) {
#pragma unroll 1
        for (int it = 0; it < ROUNDS_; ++it) {
            int64_t x = raw[0];
#pragma unroll
            for (int j = 1; j < ROUNDS_; ++j) x = j == it ? raw[j] : x;      // raw[] lives in registers: select, do not index
            const uint32_t d = ng_digits(ng_canon(x, mod));
            dst[it * ROUND] = (uint8_t)d; dst[PLANE + it * ROUND] = (uint8_t)(d >> 8); dst[2 * PLANE + it * ROUND] = (uint8_t)(d >> 16);
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < ROUNDS_; ++it) {
        const uint32_t d = ng_digits((uint32_t)raw[it]);
        dst[it * ROUND] = (uint8_t)d; dst[PLANE + it * ROUND] = (uint8_t)(d >> 8); dst[2 * PLANE + it * ROUND] = (uint8_t)(d >> 16);
    }
}

// Where the B fragments of a step live during the row tiles.  Fragments in registers have to survive the staging calls of every
// later step: the calling convention preserves 64 vector registers, and once more fragments than that crossed a call the compiler
// parked some in scratch memory and kept reloading them INSIDE the row loop, every tile, behind an s_waitcnt vmcnt(0) that also
// waits for the wave's share stores (124 k or 180 k cycles in the row-tile phase, depending on unrelated code).  So: the LAST
// step's fragments are read from the digit tile, which stays in LDS through the row tiles; the step before it (KS >= 4) from a
// stash the wave copies them to - fragment by fragment, 1 KiB each, in the side tile's memory, which is free by then; only the
// steps before those are registers (KS = 4, NT = 2: 48 of them, all within the preserved set).
// (Fetching two steps' values with one pass - one memory latency instead of two - was tried with a side tile in that memory and
// made the pass slower, 35 k against 29 k cycles: it is the bytes in flight per CU that bound it.)
template <int KS> struct NgBSource {
    static constexpr bool last_in_tile = KS >= 2, stash = KS >= 4;
    static constexpr int in_regs = stash ? KS - 2 : last_in_tile ? KS - 1 : KS;      // steps 0 .. in_regs - 1 are registers
};
template <int KS, int NT> constexpr size_t ngemm_lds_bytes() {      // A ring + digit tile + stash
    return (size_t)(KS == 8 ? 2 : 4) * (size_t)KS * 3 * 1024 + 3 * (size_t)(16 * kNgCompute * NT) * kNgRow +
           (NgBSource<KS>::stash ? (size_t)kNgCompute * NT * 3 * 1024 : 0);
}
// the CSPRNG draws d_lo .. d_lo + cd - 1 of the workgroup's batches -> digit tile.  Every prime this kernel takes (p <= 0x7F7F7F) is
// in the domain of sda-drbg-v1's PAIRED rule (modarith.hpp): one lane = one block = draws 2j AND 2j + 1 of 8 batches - 78 blocks per
// 8 batches of PSS_155_728_100 instead of 155.  A pair that straddles two 64-term steps is computed in both (each step writes the
// element that falls into its term range).
// the rare rejections of a draw pass (below 2^-18 per pair), redone from the retry stream: `rej` bit 8 * round + jj = pair (u, jj) of this
// lane's round-th block was rejected (64 bits: up to eight rounds).  Cold and not inlined - see ng_draw_pass
template <int WGB>
__device__ __noinline__ void ng_draw_fixup(uint8_t* Bt, DrbgKey key, uint64_t stream, uint64_t b0, uint32_t k, uint32_t t, uint32_t t_lo,
                                           uint32_t d_lo, uint32_t cd, uint64_t m, uint64_t thr2, uint64_t rej) {
    ng_lptr B = (ng_lptr)Bt;
    const uint32_t d_hi = d_lo + cd, t2 = (t + 1u) >> 1, j_lo = d_lo >> 1, cp = ((d_hi - 1u) >> 1) - j_lo + 1u;
    for (uint32_t bit = 0; bit < 64u; ++bit) {
        if (!((rej >> bit) & 1ull)) continue;
        const uint32_t u = threadIdx.x + (bit >> 3) * (uint32_t)kNgWorkers, jj = bit & 7u;
        const uint32_t nb = u / cp, j = j_lo + (u - nb * cp);
        const uint64_t pr = f_drbg_retry_pair<20>(key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7], stream,
                                                  (b0 + 8u * nb + jj) * (uint64_t)t2 + j, m, thr2);
        const uint32_t i0 = 2u * j, i1 = i0 + 1u;
        if (i0 >= d_lo) ng_put(B, WGB, 8u * nb + jj, k + i0 - t_lo, ng_digits((uint32_t)pr));
        if (i1 < d_hi) ng_put(B, WGB, 8u * nb + jj, k + i1 - t_lo, ng_digits((uint32_t)(pr >> 32)));
    }
}

template <int WGB>
__device__ __noinline__ void ng_draw_pass(uint8_t* Bt, DrbgKey key, uint64_t stream, uint64_t b0, uint32_t k, uint32_t t, uint32_t t_lo,
                                          uint32_t d_lo, uint32_t cd, uint64_t m, uint64_t thr2) {
    ng_lptr B = (ng_lptr)Bt;
    const uint32_t kk[8] = {key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7]};
    const uint32_t d_hi = d_lo + cd, t2 = (t + 1u) >> 1, j_lo = d_lo >> 1, cp = ((d_hi - 1u) >> 1) - j_lo + 1u;   // cd >= 1
    // A rejected pair is NOT redone here: the call to the retry stream inside this loop made every value that lives across it (the
    // block's sixteen words, the key, the loop state) a callee-saved register - 27 of them saved and restored in scratch memory by
    // EVERY call of this pass, 166 KB per workgroup, and with the L2 turning over every few microseconds those bytes went to HBM and
    // back (PSS_155_728_100: 33.7 GB written per tile for 30.5 GB of shares; the same with either share map).  The rejected pairs are
    // noted in a mask (at most five blocks per lane and pass: (WGB / 8) * cp <= 64 * 33) and redone by ng_draw_fixup, called last.
    uint64_t rej = 0;
    uint32_t round = 0;
#pragma unroll 1
    for (uint32_t u = threadIdx.x; u < (uint32_t)(WGB / 8) * cp; u += kNgWorkers, ++round) {
        const uint32_t nb = u / cp, j = j_lo + (u - nb * cp);
        const uint64_t I = ((b0 >> 3) + nb) * (uint64_t)t2 + j;
        uint32_t o[16];
        chacha_block_lane<20>(kk, (uint32_t)I, (uint32_t)(I >> 32), (uint32_t)stream, (uint32_t)(stream >> 32) & 0xFFFFFFu, o);
        const uint32_t i0 = 2u * j, i1 = i0 + 1u;
        const bool w0 = i0 >= d_lo, w1 = i1 < d_hi;                            // (i0 < d_hi and i1 >= d_lo hold by construction)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int cc = jj >> 1, e = jj & 1;
            const uint64_t xw = ((uint64_t)o[8 * e + cc] << 32) | o[8 * e + 4 + cc];
            uint32_t ra, rb;
            if (!lemire_pair(xw, (uint32_t)m, thr2, ra, rb)) rej |= 1ull << (8u * round + (uint32_t)jj);
            if (w0) ng_put(B, WGB, 8u * nb + jj, k + i0 - t_lo, ng_digits(ra));
            if (w1) ng_put(B, WGB, 8u * nb + jj, k + i1 - t_lo, ng_digits(rb));
        }
    }
    if (__builtin_expect(rej != 0, 0)) ng_draw_fixup<WGB>(Bt, key, stream, b0, k, t, t_lo, d_lo, cd, m, thr2, rej);
}

// systematic share map: draw i of a batch IS its share i.  The step's draws are read back from the tile (three digits ->
// canonical value) with lane = batch, so that a row leaves in 512-byte pieces (a lane that stored its own block's eight values
// wrote 64 rows per instruction: 12.2 -> 16.0 ms per 500-participant tile, measured)
template <int WGB>
__device__ __noinline__ void ng_direct_pass(const uint8_t* Bt, int64_t* op, size_t stride_clerk, uint64_t b0, uint64_t batches, uint32_t k,
                                            uint32_t t_lo, uint32_t d_lo, uint32_t d_hi) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // four terms per tile read (one dword per digit plane): the term offsets of the step's draws, rounded out to multiples of 4
    const uint32_t o_lo = (k + d_lo - t_lo) & ~3u, o_hi = k + d_hi - t_lo;                   // byte offsets in a tile row, o_hi <= 64
    for (uint32_t bg = 0; bg < (uint32_t)(WGB / 64); ++bg) {
        const uint32_t bl = 64u * bg + lane;
        const uint64_t b = b0 + bl;
        const __attribute__((address_space(3))) uint8_t* q = (const __attribute__((address_space(3))) uint8_t*)Bt + bl * kNgRow;
        typedef const __attribute__((address_space(3))) uint32_t* lw;
        for (uint32_t o = o_lo + 4u * wave; o < o_hi; o += 4u * (uint32_t)kNgCompute) {
            const uint32_t w0 = *(lw)(q + o);
            const uint32_t w1 = *(lw)(q + WGB * kNgRow + o);
            const uint32_t w2 = *(lw)(q + 2 * WGB * kNgRow + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t v = (uint32_t)((int32_t)(int8_t)(w0 >> (8 * j)) + 256 * (int32_t)(int8_t)(w1 >> (8 * j)) +
                                              65536 * (int32_t)(int8_t)(w2 >> (8 * j)));        // the digits of the canonical value itself
                const uint32_t term = t_lo + o + (uint32_t)j;                                    // draw term - k = share row
                if (b < batches && term >= k + d_lo && term < k + d_hi)
                    __builtin_nontemporal_store((long long)v, (__attribute__((address_space(1))) long long*)(op + (size_t)(term - k) * stride_clerk + b));
            }
        }
    }
}

// step STEP: the three passes, then the B fragments of the step (a template recursion: the fragment array must be indexed by
// constants to stay in registers)
template <int KS, int NT, int STEP>
__device__ __forceinline__ void ng_stage(ng_v4i (&bfrag)[NT][KS > 1 ? NgBSource<KS>::in_regs : 1][3], uint8_t* Bt, uint8_t* Side, const GenLayout& L, const ModParams& mod, const DrbgKey& key,
                                         const NGemmPlan& P, const int64_t* sp, const int64_t* rp, int64_t* op, uint64_t stream, uint64_t b0,
                                         uint64_t batches
#ifdef NG_TIMING
                                         , uint64_t (&ng_tm)[4]
#endif
                                         ) {
    constexpr int WB = 16 * NT, WGB = kNgCompute * WB;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, col = lane & 15u, g = lane >> 4;
    const uint32_t k = P.k, t = P.t, kt = k + t, t_lo = 64u * STEP;
    const bool worker = wave < (uint32_t)kNgCompute;
    const bool loads = t_lo < k || rp != nullptr || t_lo + 64u > kt;         // uniform: something besides CSPRNG draws in this step
    const bool draws_here = !rp && t_lo + 64u > k && t_lo < kt;              // draws d_lo .. d_hi - 1 fall into this step
    const uint32_t d_lo = t_lo > k ? t_lo - k : 0u, d_hi = draws_here ? (t_lo + 64u < kt ? t_lo + 64u : kt) - k : d_lo;
#ifdef NG_TIMING
    const uint64_t tq0 = __builtin_readcyclecounter();
#endif
    if (worker) {
        if (loads) ng_load_pass<WGB>(Bt, sp, rp, L.len, batches, b0, k, t, t_lo, mod.m, mod.mu);
#ifdef NG_TIMING
        ng_tm[0] += __builtin_readcyclecounter() - tq0;
#endif
        if (draws_here) ng_draw_pass<WGB>(Bt, key, stream, b0, k, t, t_lo, d_lo, d_hi - d_lo, mod.m, mod.lemire_thr2);
    }
#ifdef NG_TIMING
    const uint64_t tq1 = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef NG_TIMING
    const uint64_t tq2 = __builtin_readcyclecounter();
    ng_tm[1] += tq1 - tq0; ng_tm[2] += tq2 - tq1;
#endif
    if (worker) {
        if (draws_here && L.direct_rows) ng_direct_pass<WGB>(Bt, op, L.out_stride_clerk, b0, batches, k, t_lo, d_lo, d_hi);
#ifdef NG_TIMING
        ng_tm[3] += __builtin_readcyclecounter() - tq2;
#endif
        // column col of batch tile nt = batch NT col + nt of the wave (see the row tiles' stores)
        if constexpr (STEP < NgBSource<KS>::in_regs || (NgBSource<KS>::stash && STEP == KS - 2)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int lb = 0; lb < 3; ++lb) {
                    const ng_v4i f = *reinterpret_cast<const ng_v4i*>(Bt + ((size_t)lb * WGB + wave * WB + NT * col + nt) * kNgRow + 16 * g);
                    if constexpr (STEP < NgBSource<KS>::in_regs) bfrag[nt][STEP][lb] = f;
                    else *reinterpret_cast<ng_v4i*>(Side + ((size_t)(wave * NT + nt) * 3 + lb) * 1024 + lane * 16) = f;     // the wave's stash
                }
        }
    }
    __syncthreads();
#ifdef NG_TIMING
    if constexpr (STEP + 1 < KS) ng_stage<KS, NT, STEP + 1>(bfrag, Bt, Side, L, mod, key, P, sp, rp, op, stream, b0, batches, ng_tm);
#else
    if constexpr (STEP + 1 < KS) ng_stage<KS, NT, STEP + 1>(bfrag, Bt, Side, L, mod, key, P, sp, rp, op, stream, b0, batches);
#endif
}

// ---- dual-role launch (DESIGN.md 5 "Schedules"): position c * period of the grid is the c-th clerk-sum workgroup of the PREVIOUS tile
// (two items of 512 columns x one job x one row split, one per half of the compute waves; exact 128-bit sums, carry-propagating
// atomics when the rows are split), every other position a share-generation workgroup of this tile.  The clerk sum is bound by HBM
// and the share generation by the SIMDs' issue slots: a CU in the clerk role streams while its neighbours multiply.
struct NgFuse {
    uint64_t* acc_lo; int64_t* acc_hi; const int64_t* prev;
    size_t job_stride, n_rows, row_stride, dimension, rows_per_split;
    uint32_t col_blocks, jobs, splits, pad;
    uint64_t n_gen, n_comb, n_comb_wg, period;                    // n_comb items in n_comb_wg workgroups; 0: share generation only
    uint64_t n_early;                                             // clerk workgroups 0 .. n_early - 1 sit at positions c * period among the share-generation
                                                                  // workgroups, the others FOLLOW the last share-generation workgroup (they fill its tail)
    uint32_t items_per_wg;                                        // clerk items one clerk workgroup sums (even: one per half of the compute waves at a time)
    uint32_t pad2;
    // clerk WAVES (round 6, the default): every share-generation workgroup carries kNgClerkWaves waves that sum the PREVIOUS tile while
    // the compute waves multiply - items of 128 columns x all rows of one job, cw_per_slot consecutive items per (workgroup, clerk
    // wave) slot; what a slot did not get to is recorded in cw_progress (item, row) and summed by ngemm_clerk_rest_kernel afterwards
    uint64_t cw_items, cw_per_slot;
    uint64_t* cw_progress;                                        // [slots][2]
    uint32_t cw_col_blocks, cw_pad;
};
static constexpr int kNgClerkUnroll = 8;      // row loads in flight per lane in the clerk role (4: 14.9, 8: 14.6, 16: 14.9, 32: 15.9 ms per tile)
static constexpr int kNgClerkUnrollShort = 20; // ... for items of a few rows (40 participants of PSS_155_19682_100: two rounds of loads instead of five)

template <int KS> struct NgRing { static constexpr int depth = KS == 8 ? 2 : 4; };   // LDS slots of A tiles (a ring of 7 for KS = 4 changed nothing, also not for the 15 MB matrix of n = 19682)

// The store hazard of gfx950 made safe IN THE SOURCE (it used to rest on the compiler's choice of operands alone): a buffer store
// whose data registers are written again too early can leave with the NEW value (tools/microbench_store_war.hip: 2.5 % of 16-byte
// stores at 0 wait states, none at 1; see finish_whole below).  The empty-bodied wait takes the stored registers as INPUTS: they
// stay live and unwritten until it has executed, whatever the register allocator and the scheduler do around it - s_nop 1 = two wait
// states, one more than the measured need.  tests/test_ngemm_isa.py (and __graft_entry__.build()) check the assembly for it.
__device__ __forceinline__ void ng_store_guard(ng_v4u v) { asm volatile("s_nop 1" ::"v"(v)); }
__device__ __forceinline__ void ng_store_guard(ng_v2u v) { asm volatile("s_nop 1" ::"v"(v)); }

// ---- clerk waves (round 6) ---------------------------------------------------------------------------------------------------------
// The dual-role GRID gives whole CUs to the clerk sum, and a CU in that role streams 27 - 45 GB/s (one workgroup of this kernel's
// size per CU: 512 lanes of loads in flight) while its matrix cores idle; the two launches run back to back took 10.1 + 4.6 ms per
// tile of PSS_155_728_100, the dual-role grid 12.8.  A second kernel on another stream is not placed beside this one's workgroups at
// all (profiles/r06/ab_ngemm_side_waves_not_adopted.txt).  So the clerk sum rides INSIDE every share-generation workgroup: three more
// waves - the SIMDs without the loader wave have the registers and the slot - that do nothing but load rows of the previous tile
// and add them up, on ALL 256 CUs at once, in the issue slots the compute waves leave (VALU busy 0.37 - 0.46) and the HBM bandwidth
// share generation does not use (3.6 of ~6.2 TB/s).
//
// A clerk wave must arrive at every workgroup barrier (2 per staging step + 1 per row tile): a barrier STEP is its unit of work.  Step
// b: add up the R rows it asked for at step b - 2, ask for the next R rows, s_barrier - TWO register sets, so that a load has two
// barrier intervals (~4 - 5 us) to arrive: with one set (measured, profiles/r06/ab_ngemm_clerk_waves_v1.txt) the wave waited 3+ us for
// memory every step and held the whole workgroup's barrier - the share-generation kernel went from 19.9 to 30 ms on
// PSS_155_19682_100.  The wait for a set must leave the YOUNGER set in flight: s_waitcnt vmcnt(R).  The compiler cannot emit that
// for a wave that also stores (loads and stores return out of order with respect to each other, so its wait-count pass falls back
// to vmcnt(0) whenever a store may be pending - and the running sums of a finished item are stored here).  The counted wait is
// still correct: loads return in order AMONG LOADS, so "at most R operations outstanding" implies that every load older than the
// youngest R loads has returned, whatever the stores do (a pending store only makes the wait longer).  Hence the loads are issued
// and awaited in inline assembly (ng_clerk_load / ng_clerk_wait): the compiler sees values that flow from one asm statement to the
// next and has no reason to touch them in between - tests/test_ngemm_isa.py walks the assembly and fails a build in which any
// other instruction reads a register of a set between its load and its wait.  Every step issues exactly R row loads (an exhausted
// slot re-reads a valid address), so R is always a lower bound of the younger loads.
// Items = (job, 128 columns) over all rows of the tile, in a fixed order per slot; the running sums of an item are fetched with its
// first rows and written back with its last (no other wave touches them in this launch).  Whatever is left when the workgroup's
// last barrier has passed - the rest of the current item's rows and the slot's remaining items - goes to ngemm_clerk_rest_kernel
// via cw_progress.
struct NgClerkCursor {
    uint64_t it, end;          // item index (job * col_blocks + column block), end of the slot
    uint64_t job; uint32_t bx; // ... decomposed
    uint32_t r;                // first row of the next quantum
};
static constexpr int kNgClerkRows = 10;      // rows per quantum and register set (40 registers each): 40 participants = 4 quanta, 500 = 50

struct NgClerkSet {
    ll2 v[kNgClerkRows];
    ll2 al, ah;                               // the item's running sums (low words / high words of its two columns), fetched with its first quantum
};

__device__ __forceinline__ void ng_clerk_advance(NgClerkCursor& c, uint32_t n_rows, uint32_t col_blocks) {
    c.r += kNgClerkRows;
    if (c.r >= n_rows) {
        c.r = 0; ++c.it;
        if (++c.bx == col_blocks) { c.bx = 0; ++c.job; }
    }
}
// one 16-byte load the compiler does not track ("NGCL" tags it for the ISA test).  nt: the shares are read once
__device__ __forceinline__ void ng_clerk_load(ll2& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off nt ; NGCL" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ng_clerk_load_plain(ll2& dst, const void* p) {           // the running sums: cached like any data
    asm volatile("global_load_dwordx4 %0, %1, off ; NGCL" : "=v"(dst) : "v"(p) : "memory");
}
// every load of the set has returned when at most `YOUNGER` vector-memory operations are outstanding (see above); the set's registers
// are read-write operands, so that no use of them can be scheduled before this statement ("NGCW" + the registers for the ISA test)
template <int YOUNGER>
__device__ __forceinline__ void ng_clerk_wait(NgClerkSet& S) {
    static_assert(kNgClerkRows == 10, "the operand list below names ten rows");
    asm volatile("s_waitcnt vmcnt(%12) ; NGCW %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11"
                 : "+v"(S.v[0]), "+v"(S.v[1]), "+v"(S.v[2]), "+v"(S.v[3]), "+v"(S.v[4]), "+v"(S.v[5]), "+v"(S.v[6]), "+v"(S.v[7]),
                   "+v"(S.v[8]), "+v"(S.v[9]), "+v"(S.al), "+v"(S.ah)
                 : "n"(YOUNGER)
                 : "memory");
}

// `extra`: two bits per staging step - that many PAIRS of quanta more before the step's first barrier.  The staging steps that draw
// from the CSPRNG keep the compute waves busy for 5 - 10 us (two ChaCha20 blocks per lane and more) while HBM idles; a clerk wave
// that did one quantum per barrier there summed a quarter of what it sums per microsecond in the row tiles, and 12 % of a
// PSS_155_728_100 tile was left to the follow-up kernel.
__device__ __forceinline__ void ng_clerk_wave(const NgFuse& F, uint64_t slot, uint32_t lane, uint32_t n_barriers, uint32_t extra) {
    if (F.cw_items == 0) {                                           // a launch without a previous tile: the barriers, nothing else
        for (uint32_t b = 0; b < n_barriers; ++b) __builtin_amdgcn_s_barrier();
        return;
    }
    const uint32_t n_rows = (uint32_t)F.n_rows, col_blocks = F.cw_col_blocks;
    NgClerkCursor is, cs;                                            // issue side, consume side (the same walk, two quanta behind)
    is.it = slot * F.cw_per_slot < F.cw_items ? slot * F.cw_per_slot : F.cw_items;
    is.end = is.it + F.cw_per_slot < F.cw_items ? is.it + F.cw_per_slot : F.cw_items;
    is.job = is.it / col_blocks; is.bx = (uint32_t)(is.it - is.job * col_blocks); is.r = 0;
    cs = is;
    uint64_t lo0 = 0, lo1 = 0, bl0 = 0, bl1 = 0;                      // sums of the item being consumed; its running sums from memory
    int64_t hi0 = 0, hi1 = 0, bh0 = 0, bh1 = 0;
    NgClerkSet A, B;
    bool validA = false, validB = false;
    // the running sums are read and written 16 bytes (both columns of the lane) at a time: acc_lo / acc_hi + job * dimension + c0 is
    // 16-byte aligned when dimension is even (c0 is); an odd dimension is not given to the clerk waves (ngemm_launch_fused)
    auto issue = [&](NgClerkSet& S, bool& valid) {
        valid = is.it < is.end;
        size_t c0 = 2 * ((size_t)is.bx * 64 + lane);
        if (c0 + 1 >= F.dimension) c0 = 0;                           // a lane beyond the last column pair re-reads columns 0, 1 (never stored)
        const uint64_t job = valid ? is.job : 0;                     // (an exhausted slot: the same R loads from a valid address)
        const uint32_t r0 = valid ? is.r : 0;
        const int64_t* base = F.prev + job * F.job_stride + c0;
        if (valid && r0 == 0) {
            const size_t idx = job * F.dimension + c0;
            ng_clerk_load_plain(S.al, F.acc_lo + idx);
            ng_clerk_load_plain(S.ah, F.acc_hi + idx);
        }
#pragma unroll
        for (int u = 0; u < kNgClerkRows; ++u) {
            const uint32_t rr = r0 + u < n_rows ? r0 + u : n_rows - 1;                    // (clamped: never added)
            ng_clerk_load(S.v[u], base + (size_t)rr * F.row_stride);
        }
        if (valid) ng_clerk_advance(is, n_rows, col_blocks);
    };
    auto flush = [&]() {                                             // the consumed rows of the current item -> its running sums in memory
        const size_t c0 = 2 * ((size_t)cs.bx * 64 + lane);
        if (c0 + 1 >= F.dimension) return;
        const size_t idx = cs.job * F.dimension + c0;
        const uint64_t n0 = bl0 + lo0, n1 = bl1 + lo1;
        ll2 l, h;
        l.x = (long long)n0; l.y = (long long)n1;
        h.x = bh0 + hi0 + (n0 < bl0 ? 1 : 0); h.y = bh1 + hi1 + (n1 < bl1 ? 1 : 0);
        *reinterpret_cast<ll2*>(F.acc_lo + idx) = l;
        *reinterpret_cast<ll2*>(F.acc_hi + idx) = h;
    };
    auto consume = [&](NgClerkSet& S, bool& valid) {                 // (after ng_clerk_wait(S))
        if (!valid) return;
        valid = false;
        if (cs.r == 0) {
            bl0 = (uint64_t)S.al.x; bl1 = (uint64_t)S.al.y; bh0 = S.ah.x; bh1 = S.ah.y;
            lo0 = lo1 = 0; hi0 = hi1 = 0;
        }
#pragma unroll
        for (int u = 0; u < kNgClerkRows; ++u)
            if (cs.r + u < n_rows) { acc_add(lo0, hi0, S.v[u].x); acc_add(lo1, hi1, S.v[u].y); }
        if (cs.r + kNgClerkRows >= n_rows) flush();
        ng_clerk_advance(cs, n_rows, col_blocks);
    };
    // (the sets start as zeros: the first waits name registers nothing was loaded into yet)
#pragma unroll
    for (int u = 0; u < kNgClerkRows; ++u) { A.v[u] = ll2{0, 0}; B.v[u] = ll2{0, 0}; }
    A.al = A.ah = B.al = B.ah = ll2{0, 0};
    uint32_t b = 0;
    for (; b + 1 < n_barriers; b += 2) {
        // (barrier b = 2 x step is a staging step's first barrier; `extra` holds that step's two bits at position b, zeros beyond the steps)
        const uint32_t e = b < 32u ? (extra >> b) & 3u : 0u;
        if (e >= 1u) {
            ng_clerk_wait<kNgClerkRows>(A);
            consume(A, validA); issue(A, validA);
            ng_clerk_wait<kNgClerkRows>(B);
            consume(B, validB); issue(B, validB);
        }
        if (e >= 2u) {
            ng_clerk_wait<kNgClerkRows>(A);
            consume(A, validA); issue(A, validA);
            ng_clerk_wait<kNgClerkRows>(B);
            consume(B, validB); issue(B, validB);
        }
        ng_clerk_wait<kNgClerkRows>(A);                              // B's rows (asked for one step ago) stay in flight
        consume(A, validA); issue(A, validA);
        __builtin_amdgcn_s_barrier();
        ng_clerk_wait<kNgClerkRows>(B);
        consume(B, validB); issue(B, validB);
        __builtin_amdgcn_s_barrier();
    }
    if (b < n_barriers) {                                            // an odd number of barriers: A is the younger set afterwards
        ng_clerk_wait<kNgClerkRows>(A);
        consume(A, validA); issue(A, validA);
        __builtin_amdgcn_s_barrier();
        ng_clerk_wait<0>(B); consume(B, validB);
        ng_clerk_wait<0>(A); consume(A, validA);
    } else {
        ng_clerk_wait<0>(A); consume(A, validA);
        ng_clerk_wait<0>(B); consume(B, validB);
    }
    // the workgroup is done: the rows consumed of an unfinished item go to memory, the rest of the slot to the follow-up kernel
    if (cs.it < cs.end && cs.r != 0) flush();
    if (F.cw_progress && lane == 0) { F.cw_progress[2 * slot] = cs.it; F.cw_progress[2 * slot + 1] = cs.r; }
}

// what the clerk waves of a launch did not get to: one single-wave workgroup per slot, from the recorded (item, row) to the slot's end.
// The same sums in the same layout (items = 128 columns of one job), kNgClerkRest rows in flight per lane.
static constexpr int kNgClerkRest = 16;
__global__ __launch_bounds__(64) void ngemm_clerk_rest_kernel(NgFuse F, uint64_t slots) {
    const uint32_t lane = threadIdx.x, n_rows = (uint32_t)F.n_rows;
    for (uint64_t slot = blockIdx.x; slot < slots; slot += gridDim.x) {
        uint64_t it = F.cw_progress[2 * slot];
        uint32_t r0 = (uint32_t)F.cw_progress[2 * slot + 1];
        const uint64_t first = slot * F.cw_per_slot < F.cw_items ? slot * F.cw_per_slot : F.cw_items;
        const uint64_t end = first + F.cw_per_slot < F.cw_items ? first + F.cw_per_slot : F.cw_items;
        for (; it < end; ++it, r0 = 0) {
            const uint64_t job = it / F.cw_col_blocks, bx = it - job * F.cw_col_blocks;
            const size_t c0 = 2 * ((size_t)bx * 64 + lane);
            if (c0 >= F.dimension) continue;
            const bool two = c0 + 1 < F.dimension;
            const int64_t* base = F.prev + job * F.job_stride + c0;
            const size_t idx = job * F.dimension + c0;
            const uint64_t l0 = F.acc_lo[idx], l1 = two ? F.acc_lo[idx + 1] : 0;
            const int64_t h0 = F.acc_hi[idx], h1 = two ? F.acc_hi[idx + 1] : 0;
            uint64_t lo0 = 0, lo1 = 0;
            int64_t hi0 = 0, hi1 = 0;
            for (uint32_t r = r0; r < n_rows; r += kNgClerkRest) {
                ll2 v[kNgClerkRest];
#pragma unroll
                for (int u = 0; u < kNgClerkRest; ++u) {
                    const uint32_t rr = r + u < n_rows ? r + u : n_rows - 1;
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(base + (size_t)rr * F.row_stride));
                }
#pragma unroll
                for (int u = 0; u < kNgClerkRest; ++u)
                    if (r + u < n_rows) { acc_add(lo0, hi0, v[u].x); if (two) acc_add(lo1, hi1, v[u].y); }
            }
            uint64_t nl = l0 + lo0;
            F.acc_lo[idx] = nl; F.acc_hi[idx] = h0 + hi0 + (nl < l0 ? 1 : 0);
            if (two) { nl = l1 + lo1; F.acc_lo[idx + 1] = nl; F.acc_hi[idx + 1] = h1 + hi1 + (nl < l1 ? 1 : 0); }
        }
    }
}

template <int N> __device__ __forceinline__ void ng_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The loader wave's part of the row tiles, a function of its own (NOT inlined): in one build of the <8, 1> instance the compiler merged
// this loop with the compute waves' row loop (one loop, the per-tile barrier shared) and parked the loader's lane offset in scratch
// memory - reloaded every tile, behind a vmcnt(0) that also waits for the tiles in flight.  A call keeps the two loops apart.
template <int KS>
__device__ __noinline__ void ng_loader_rows(const uint8_t* A, uint8_t* Abuf, uint32_t tiles, uint32_t lane) {
    constexpr int PIECES = KS * 3, ATILE = PIECES * 1024, DEPTH = NgRing<KS>::depth;
    auto issue_tile = [&](uint32_t tile, uint32_t slot) {
        const uint8_t* src = A + (size_t)tile * ATILE + lane * 16;
        uint8_t* dst = Abuf + slot * ATILE;
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    };
    uint32_t slot = DEPTH - 1;                                      // slot of tile rt + DEPTH - 1
    for (uint32_t rt = 0; rt < tiles; ++rt) {
        const uint32_t nxt = rt + DEPTH - 1;
        if (nxt < tiles) {
            issue_tile(nxt, slot);                                  // the slot tile rt - 1 was read from (free since the last barrier)
            ng_wait_vm<PIECES * (DEPTH - 2) < 63 ? PIECES * (DEPTH - 2) : 0>();     // tile rt + 1 has landed
        } else {
            ng_wait_vm<0>();
        }
        slot = slot + 1 == (uint32_t)DEPTH ? 0u : slot + 1;
        __builtin_amdgcn_s_barrier();
    }
}

template <int KS, int NT>
// 168 registers: three waves per SIMD, i.e. the ten waves of two workgroups on a CU's four SIMDs
__global__ __launch_bounds__(kNgThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void packed_gen_ngemm_kernel(GenLayout L, ModParams mod, DrbgKey key, NGemmPlan P,
                                                                        uint64_t chunks, uint64_t batches, NgFuse F) {
    constexpr int WB = 16 * NT, WGB = kNgCompute * WB;                       // batches per wave / per workgroup
    constexpr int PIECES = KS * 3, ATILE = PIECES * 1024;           // one row tile: PIECES fragments of 1 KiB
    constexpr int DEPTH = NgRing<KS>::depth;
    extern __shared__ __align__(16) uint8_t ng_lds[];
    uint8_t* Abuf = ng_lds;                                          // [DEPTH][ATILE]
    uint8_t* Bt = ng_lds + DEPTH * ATILE;                            // [3][WGB][kNgRow]
    uint8_t* Side = Bt + 3 * WGB * kNgRow;                           // [wave][NT][3][1 KiB]: the stash of step KS - 2's B fragments (KS >= 4)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, col = lane & 15u, g = lane >> 4;
    uint64_t item = blockIdx.x;
    // grid position -> clerk workgroup index (role 1) or share-generation item (role 0); role 2: a surplus position
    auto role_at = [&](uint64_t pos, uint64_t& idx) -> int {
        if (!F.n_comb_wg) { idx = pos; return pos < (uint64_t)gridDim.x ? 0 : 2; }
        const uint64_t mixed = F.n_gen + F.n_early;                 // positions below this: share generation with the early clerk positions in between
        if (pos >= mixed) { idx = F.n_early + (pos - mixed); return idx < F.n_comb_wg ? 1 : 2; }
        const uint64_t q = pos / F.period, rem = pos - q * F.period;
        if (rem == 0 && q < F.n_early) { idx = q; return 1; }
        const uint64_t before = q + (rem ? 1 : 0);                   // clerk positions below this one
        idx = pos - (before < F.n_early ? before : F.n_early);
        return idx < F.n_gen ? 0 : 2;
    };
    {
        uint64_t idx;
        const int role = role_at(item, idx);
        if (role == 2) return;
        if (role == 1) {
            // a clerk workgroup sums items_per_wg consecutive items (adjacent column blocks of one job, then the next job's): two at a
            // time, threads 0-255 the even one and threads 256-511 the odd one.  (One pair per workgroup, as the launch had it for
            // shapes with fewer clerk items than share-generation workgroups, made 206 k clerk workgroups of PSS_155_19682_100's 413 k
            // items - all of them in front of the 1640 share-generation workgroups: the two roles ran one after the other.)
            if (tid < 512u) {
                const uint64_t first = idx * F.items_per_wg + (tid >> 8);
                for (uint32_t j = 0; j < F.items_per_wg; j += 2) {
                    const uint64_t it = first + j;
                    if (it >= F.n_comb) break;
                    const uint64_t bx = it % F.col_blocks, rest = it / F.col_blocks;
                    if (F.rows_per_split % kNgClerkUnrollShort == 0 && F.rows_per_split <= 4 * kNgClerkUnrollShort)
                        combine_pair<true, kNgClerkUnrollShort>(F.acc_lo, F.acc_hi, F.prev, F.job_stride, F.n_rows, F.row_stride, F.dimension,
                                                                F.rows_per_split, F.splits > 1, bx * 256u + (tid & 255u), rest % F.jobs, rest / F.jobs);
                    else
                        combine_pair<true, kNgClerkUnroll>(F.acc_lo, F.acc_hi, F.prev, F.job_stride, F.n_rows, F.row_stride, F.dimension,
                                                           F.rows_per_split, F.splits > 1, bx * 256u + (tid & 255u), rest % F.jobs, rest / F.jobs);
                }
            }
            return;
        }
        item = idx;
    }
    if (wave > (uint32_t)kNgCompute) {                              // a clerk wave: every barrier of the workgroup, nothing else of it
        const uint32_t cw = __builtin_amdgcn_readfirstlane(wave) - (uint32_t)kNgCompute - 1u;
        // pairs of quanta more in the staging steps that draw (no injected randomness): two for 48 draws and more, one for 24
        uint32_t extra = 0;
        if (!L.rand) {
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const uint32_t lo = 64u * st > P.k ? 64u * st : P.k, hi = 64u * st + 64u < P.k + P.t ? 64u * st + 64u : P.k + P.t;
                const uint32_t draws = hi > lo ? hi - lo : 0u;
                extra |= (draws >= 48u ? 2u : draws >= 24u ? 1u : 0u) << (2 * st);      // two bits at the step's first barrier index, 2 st
            }
        }
        ng_clerk_wave(F, item * (uint64_t)kNgClerkWaves + cw, lane, 2u * (uint32_t)KS + P.row_tiles, extra);
        return;
    }
    const uint64_t p = item / chunks, chunk = item - p * chunks;
    const uint64_t b0 = chunk * WGB;
    const int64_t* sp = L.secrets + p * L.secrets_stride;
    const int64_t* rp = L.rand ? L.rand + p * L.rand_stride : nullptr;
    const uint64_t stream = L.first_participant + p;
    const uint32_t tiles = P.row_tiles;
    // The last wave is the LOADER: it only moves A tiles global -> LDS (global_load_lds, 1 KiB per instruction, no registers) and
    // keeps the barriers.  The compute waves never load in the row loop, so nothing ever waits for their share stores: vmcnt is
    // an in-order counter, and with the loads in the compute waves every tile waited for the previous tile's non-temporal
    // stores to be acknowledged by HBM (52 % of the wave cycles in wait states, measured).
    const bool loader = wave == (uint32_t)kNgCompute;
    auto issue_tile = [&](uint32_t tile, uint32_t slot) {
        const uint8_t* src = P.A + (size_t)tile * ATILE + lane * 16;
        uint8_t* dst = Abuf + slot * ATILE;
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    };
    if (loader) {
        for (uint32_t tile = 0; tile < (uint32_t)(DEPTH - 1) && tile < tiles; ++tile) issue_tile(tile, tile);
        ng_wait_vm<0>();                                            // landed before the first of the staging phase's barriers
    }

    // ---- values -> B fragments, one 64-term step at a time -------------------------------------------------------------------
#if defined(NG_TIMING) || defined(NG_PHASES)
    const uint64_t ng_t0 = __builtin_readcyclecounter();
    const uint64_t ng_w0 = wall_clock64();
#endif
    ng_v4i bfrag[NT][KS > 1 ? NgBSource<KS>::in_regs : 1][3];
    int64_t* op = L.out + p * L.out_stride_participant;
#ifdef NG_TIMING
    uint64_t ng_tm[4] = {0, 0, 0, 0};
    ng_stage<KS, NT, 0>(bfrag, Bt, Side, L, mod, key, P, sp, rp, op, stream, b0, batches, ng_tm);
    const uint64_t ng_t1 = __builtin_readcyclecounter();
#else
    ng_stage<KS, NT, 0>(bfrag, Bt, Side, L, mod, key, P, sp, rp, op, stream, b0, batches);
#endif
#ifdef NG_PHASES
    const uint64_t ng_t1 = __builtin_readcyclecounter();            // (the two phases only: the per-pass timers of NG_TIMING cost registers)
#endif

    // ---- the row tiles ---------------------------------------------------------------------------------------------------
    if (loader) {
        ng_loader_rows<KS>(P.A, Abuf, tiles, lane);
        return;
    }
    // this lane's output rows: 4 g + i of every tile; its batch columns: bl + nt, ADJACENT (column col of batch tile nt is batch
    // NT col + nt of the wave), so that a lane stores two shares of a row with one 16-byte instruction: the CU's address path
    // takes a wave's store lane by lane, and 4 NT instructions of 8 bytes per lane and tile kept it busy for a fifth of the
    // row-tile phase (measured with the stores compiled out: 189k -> 147k cycles per workgroup of PSS_155_728_100)
    const uint32_t bl = wave * WB + (uint32_t)NT * col;            // first of this lane's batch columns, inside the workgroup
    // (systematic share map: the matrix has the rows direct_rows .. n - 1, rows 0 .. direct_rows - 1 were the draws)
    // Addresses: a UNIFORM row pointer (scalar registers, stepped by 16 rows per tile) plus ONE 32-bit lane offset (column and
    // row 4 g) - row pointers per lane were 10 vector registers of a loop that has none to spare (see finish()).
    // (the row pointer is wave-uniform by construction; saying so keeps it in scalar registers - in one build of the <8, 1> instance
    // it had ended up in vector registers and every masked store became a readfirstlane loop around its descriptor)
    const uint64_t obase_v = reinterpret_cast<uint64_t>(op + b0 + (size_t)(rp ? 0u : L.direct_rows) * L.out_stride_clerk);
    const uint32_t obase_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(obase_v >> 32));       // (the builtin returns a SIGNED int:
    const uint32_t obase_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)obase_v);               //  widen each half as unsigned)
    const char* obase = reinterpret_cast<const char*>(((uint64_t)obase_hi << 32) | (uint64_t)obase_lo);
    const size_t row_bytes = L.out_stride_clerk * sizeof(int64_t);
    const uint32_t loff = bl * 8u + 4u * g * (uint32_t)row_bytes;  // (the fast path asks for 16 rows below 4 GiB)
    const bool late = wave >= (uint32_t)(kNgCompute / 2);
    constexpr int KREG = NgBSource<KS>::in_regs, NLDS = KS - KREG;  // steps in registers / in LDS (0, 1 or 2)
    // The order in which a row tile's matrix instructions take the steps: the LDS steps BETWEEN register steps (0, KS - 2, 1,
    // KS - 1, 2, ...), so that ONE buffer of 3 NT fragments serves both - the second LDS step is fetched into it while a register
    // step multiplies.  All NT batch tiles of the wave are in flight (each A fragment is read from LDS once per row tile).
    auto step_at = [](int pos) constexpr {
        if (NLDS == 2) return pos == 0 ? 0 : pos == 1 ? KS - 2 : pos == 2 ? 1 : pos == 3 ? KS - 1 : pos - 2;
        return pos;                                                 // (NLDS == 1: the last step is the LDS one and comes last)
    };
    ng_v4i acc[NT][5];
    // (tests/test_ngemm_isa.py fails a build whose row loop touches scratch memory.)
    auto products = [&](uint32_t slot_) {
        constexpr int GROUPS = KS * 3;
        const uint8_t* Acur = Abuf + slot_ * ATILE + (size_t)lane * 16;
        uint32_t bt_off = (wave * WB + NT * col) * kNgRow + 16u * g;           // this lane's rows of the digit tile (plane 0, batch tile 0)
        uint32_t st_off = wave * NT * 3u * 1024u + lane * 16u;                 // the wave's stash
        asm volatile("" : "+v"(bt_off), "+v"(st_off));              // (opaque: the reads stay in the loop, they are not hoisted into registers)
        ng_v4i bl[NT][3];                                           // the fragments of the LDS step at hand
        auto fetch_b = [&](int ks) {                                // ks >= KREG, a constant after unrolling
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int lb = 0; lb < 3; ++lb)
                    bl[nt][lb] = ks == KS - 1 ? *reinterpret_cast<const ng_v4i*>(Bt + ((size_t)lb * WGB + nt) * kNgRow + bt_off)
                                              : *reinterpret_cast<const ng_v4i*>(Side + (size_t)(nt * 3 + lb) * 1024 + st_off);
        };
        auto fetch_a = [&](int gi) { return *reinterpret_cast<const ng_v4i*>(Acur + (size_t)(step_at(gi / 3) * 3 + gi % 3) * 1024); };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[nt][c] = ng_v4i{0, 0, 0, 0};
        // A fragments AHEAD groups ahead of their use (three; two for the single-step instance, whose 4 x 5 accumulators leave the
        // loop no register to spare); the first LDS step's B fragments at the start, the second one's when the first has been
        // multiplied (position 2 of the order, three groups ahead of position 3)
        constexpr int AHEAD = KS == 1 ? 2 : 3;
        ng_v4i a[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) a[j] = fetch_a(j < GROUPS ? j : 0);
        if constexpr (NLDS >= 1) fetch_b(NLDS == 2 ? KS - 2 : KS - 1);
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int ks = step_at(gi / 3), la = gi % 3;
            ng_v4i an = a[AHEAD - 1];
            if (gi + AHEAD < GROUPS) an = fetch_a(gi + AHEAD);
            if (NLDS == 2 && gi == 6) fetch_b(KS - 1);
#pragma unroll
            for (int lb = 0; lb < 3; ++lb)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], ks >= KREG ? bl[nt][lb] : bfrag[nt][ks < KREG ? ks : 0][lb], acc[nt][la + lb], 0, 0, 0);
#pragma unroll
            for (int j = 0; j + 1 < AHEAD; ++j) a[j] = a[j + 1];
            a[AHEAD - 1] = an;
        }
    };
    // uniform: every batch column of the workgroup exists, 16-byte stores are aligned, a tile's 16 rows span less than 4 GiB
    const bool near = row_bytes < (1ull << 28);                     // 16 rows of a tile within the 4 GiB a buffer descriptor spans
    const bool whole = near && b0 + WGB <= batches && ((reinterpret_cast<uintptr_t>(op) | row_bytes) & 15u) == 0;
    // (the five constants live in vector registers only while a tile is reduced: moved there by finish(), instead of occupying
    // five of the loop's registers through the matrix instructions)
    // Columns 3 and 4 share one multiplication: c_4 = 256 c_3 (mod p), and C_3 + 256 C_4 fits 32 bits - column 4 holds only products
    // of TOP digits, at most 65 for a centred matrix entry (|m| < 2^22) and 127 for a value: 512 x 65 x 128 x 256 + |C_3| < 2^31
    // (tests/test_ngemm_model.py).  One full-rate shift-add instead of a quarter-rate 64-bit multiply-add per share.
    int32_t c0, c1, c2, c3;
    auto reduce1 = [&](int nt, int i) {
        const int32_t top = acc[nt][3][i] + (int32_t)((uint32_t)acc[nt][4][i] << 8);
        int64_t S = ng_mad(c0, acc[nt][0][i], 0);
        S = ng_mad(c1, acc[nt][1][i], S);
        S = ng_mad(c2, acc[nt][2][i], S);
        S = ng_mad(c3, top, S);
        return ng_redc(S, P.np);
    };
    // shares = rows 16 rt + 4 g + i of tile rt, canonical, clerk-major (batched.rs:46-48).
    // A WHOLE tile (all but a participant's last chunk / the last row tile): no masks, a lane's two adjacent batch columns leave as
    // one 16-byte store as soon as they are reduced.  Buffer stores: the tile's row pointer in a scalar descriptor, row i as the
    // scalar offset, the lane offset in ONE vector register.
    // The row offset goes into the DESCRIPTOR (one per row: two scalar adds), the scalar offset stays 0.  With the row in an SGPR soffset the compiler puts no
    // wait state between buffer_store_dwordx4 and a write to its data registers (its rule: "no hazard when soffset is a register"),
    // and on this chip the store then now and then reads the NEW value: one launch in ten left with row i + 1's first share in row
    // i, 16 lanes of one store, only in the waves that reduce at full speed.  tools/microbench_store_war.hip: 2.5 % of such stores
    // with 0 wait states, none with 1; global_store_dwordx4 needs 2 (the compiler inserts them); 8-byte stores need none
    // (profiles/r05/microbench_store_war.txt).  tests/test_ngemm_isa.py rejects a 16-byte buffer store with an SGPR soffset, and
    // since round 6 every buffer store is followed by ng_store_guard(): the hazard no longer depends on what the compiler hoists.
    // The even columns' shares are reduced first, into registers of their own (the arrangement the kernel shipped with): the data
    // of a store is then not written again for a whole reduction either.
    auto finish_whole = [&](uint32_t rt) {
        char* tbase = const_cast<char*>(obase) + (size_t)rt * 16u * row_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tbase, 0, 0xFFFFFFFFu, 0x00020000);
        if constexpr (NT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ng_v2u v = {reduce1(0, i), 0u};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, loff, (uint32_t)i * (uint32_t)row_bytes, 2);       // aux 2: non-temporal
                ng_store_guard(v);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; nt += 2) {                    // a lane's columns nt, nt + 1
                uint32_t even[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) even[i] = reduce1(nt, i);
#pragma unroll
                for (int i = 0; i < 4; ++i) {                       // (row i: its own descriptor - scalar adds, no vector register)
                    const ng_v4u v = {even[i], 0u, reduce1(nt + 1, i), 0u};
                    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(tbase + (size_t)i * row_bytes, 0, 0xFFFFFFFFu, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsi, loff + 8 * nt, 0, 2);
                    ng_store_guard(v);
                }
            }
        }
    };
    // any tile: one share at a time under its mask - the same addressing, so that this path keeps two registers (lim, 4 g) alive
    // through the loop and not a set of row pointers
    const uint32_t lim = b0 + bl >= batches ? 0u : (batches - b0 - bl > (uint64_t)NT ? (uint32_t)NT : (uint32_t)(batches - b0 - bl));   // this lane's batch columns that exist
    auto finish_masked = [&](uint32_t rt) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(obase) + (size_t)rt * 16u * row_bytes, 0, 0xFFFFFFFFu, 0x00020000);
        const uint32_t row0 = 16u * rt + 4u * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ng_v2u v = {reduce1(nt, i), 0u};
                if ((uint32_t)nt < lim && row0 + (uint32_t)i < P.n) {
                    __builtin_amdgcn_raw_buffer_store_b64(v, rs, loff + 8 * nt, (uint32_t)i * (uint32_t)row_bytes, 2);
                    ng_store_guard(v);
                }
            }
    };
    // clerk rows 256 MiB or more apart (33 M batches of one tile): plain 64-bit addresses, computed here and now (the inputs pass
    // through an empty asm so that nothing of this is kept in registers through the loop)
    auto finish_far = [&](uint32_t rt) {
        uint32_t gg = g, bb = bl;
        asm volatile("" : "+v"(gg), "+v"(bb));
        const uint32_t row0 = 16u * rt + 4u * gg;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t v = reduce1(nt, i);
                if ((uint32_t)nt < lim && row0 + (uint32_t)i < P.n)
                    __builtin_nontemporal_store((long long)v, reinterpret_cast<long long*>(const_cast<char*>(obase) + (size_t)(row0 + (uint32_t)i) * row_bytes) + bb + nt);
            }
    };
    // whole_tiles: the row tiles this workgroup stores without masks
    const uint32_t whole_tiles = whole ? (P.n / 16u < tiles ? P.n / 16u : tiles) : 0u;
    auto finish = [&](uint32_t rt) {
        c0 = ng_pin_vgpr(P.c[0]); c1 = ng_pin_vgpr(P.c[1]); c2 = ng_pin_vgpr(P.c[2]); c3 = ng_pin_vgpr(P.c[3]);
        if (rt < whole_tiles) finish_whole(rt); else if (near) finish_masked(rt); else finish_far(rt);
        __builtin_amdgcn_sched_barrier(0);                          // the next tile's matrix instructions stay behind this reduction
    };
    // (Rounds 4 - 5 touched the NEXT workgroup's secrets ten tiles before this one ended - an L2 prefetch that bought 21 k cycles of
    // staging per workgroup and fetched every secret twice (9.6 GB for 4.2 GB of secrets per 500-participant tile).  With the clerk
    // waves streaming the previous tile through the same L2 it no longer pays: 12.6 ms per tile without it, 12.7 - 13.0 with it at any
    // distance (profiles/r06/ab_ngemm_prefetch.txt) - removed.)
    // The two compute waves of a SIMD run half a period apart: waves 0-3 multiply tile r and THEN reduce and store it, waves 4-7
    // first reduce and store tile r - 1 and then multiply tile r - so one wave's products run beside the other's vector work.
    // (With every wave in the same order the per-tile barrier keeps all of them in lockstep: all on the matrix cores, then all on
    // the vector ALUs - 3700 cycles per tile where 2300 are matrix-core time, measured.)
    uint32_t slot = 0;
    for (uint32_t rt = 0; rt < tiles; ++rt) {
        if (late && rt) finish(rt - 1);
        products(slot);
        slot = slot + 1 == (uint32_t)DEPTH ? 0u : slot + 1;
        if (!late) finish(rt);
        __syncthreads();                                            // tile rt + 1 is in LDS (the loader waited for it); slot of tile rt is free
    }
    if (late && tiles) finish(tiles - 1);                           // (a systematic plan with n == t has no matrix rows at all)
#ifdef NG_PHASES
    if (tid == 0) {
        int64_t* o = L.out + p * L.out_stride_participant + b0;
        o[0] = (int64_t)(ng_t1 - ng_t0);
        o[1] = (int64_t)(__builtin_readcyclecounter() - ng_t1);
        o[6] = (int64_t)ng_w0; o[7] = (int64_t)wall_clock64();
    }
#endif
#ifdef NG_TIMING
    if (tid == 0) {                                                 // timing build only: overwrites two shares with cycle counts
        int64_t* o = L.out + p * L.out_stride_participant + b0;
        o[0] = (int64_t)(ng_t1 - ng_t0);
        o[1] = (int64_t)(__builtin_readcyclecounter() - ng_t1);
        o[2] = (int64_t)ng_tm[0]; o[3] = (int64_t)ng_tm[1]; o[4] = (int64_t)ng_tm[2]; o[5] = (int64_t)ng_tm[3];
        o[6] = (int64_t)ng_w0; o[7] = (int64_t)wall_clock64();        // 100 MHz: the workgroup's start and end on the device's constant clock
    }
    if (lane == 0) {                                                // every compute wave's own staging passes
        int64_t* o = L.out + p * L.out_stride_participant + b0 + 8 + 4 * wave;
        o[0] = (int64_t)ng_tm[0]; o[1] = (int64_t)ng_tm[1]; o[2] = (int64_t)ng_tm[2]; o[3] = (int64_t)ng_tm[3];
    }
#endif
}

// The kernel's prime bound and the paired rule's domain are ONE constant: ng_draw_pass draws with the paired rule unconditionally,
// and every family must draw the same stream for the same key (the fft, l31 and narrow kernels switch on drbg_paired(m)).
static constexpr uint64_t kNgPrimeMax = 0x7F7F7Full;                 // v + 0x808080 < 2^24 for every residue: three digits
static_assert(kNgPrimeMax == kDrbgPairedMax, "the limb GEMM draws with the paired rule for every prime it accepts");
bool packed_ngemm_path_available(uint32_t k, uint32_t t, uint64_t p) {
    return k >= 1 && k + t >= 1 && k + t <= 512 && p <= kNgPrimeMax && drbg_paired(p);
}
uint32_t packed_ngemm_steps(uint32_t k, uint32_t t) {                // 64-term steps the compiled instances provide: 1, 2, 4, 8
    const uint32_t need = (k + t + 63) / 64;
    return need <= 1 ? 1u : need <= 2 ? 2u : need <= 4 ? 4u : 8u;
}

size_t ngemm_tile_bytes(uint32_t ks) { return (size_t)ks * 3 * 1024; }     // one row tile of A fragments

// CUs of the current device (one workgroup per CU: the stride between a workgroup and the next one on its CU); 0 if unknown
static uint32_t ng_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n > 0 ? (uint32_t)n : 0u;
}

template <int KS, int NT>
static hipError_t ngemm_launch(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, hipStream_t s) {
    constexpr int WGB = 16 * kNgCompute * NT;
    const uint64_t batches = (L.len + P.k - 1) / P.k;
    const uint64_t chunks = (batches + WGB - 1) / WGB;
    if (chunks * L.participants == 0) return hipSuccess;
    const size_t lds = ngemm_lds_bytes<KS, NT>();
    auto kern = packed_gen_ngemm_kernel<KS, NT>;
    if (lds > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    const uint64_t max_blocks = 0x7FFFFFFFull;
    uint64_t per = max_blocks / chunks;
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
        note_kernel("packed_gen_ngemm_kernel<%d, %d>", KS, NT);
        NgFuse F{};
        kern<<<dim3((unsigned)(chunks * cnt)), dim3(kNgThreads), lds, s>>>(S, mod, key, P, chunks, batches, F);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_packed_generate_ngemm(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, hipStream_t s) {
    switch (P.ks) {
        case 1: return ngemm_launch<1, 4>(L, mod, key, P, s);
        case 2: return ngemm_launch<2, 2>(L, mod, key, P, s);
        case 4: return ngemm_launch<4, 2>(L, mod, key, P, s);
        case 8: return ngemm_launch<8, 1>(L, mod, key, P, s);
        default: return hipErrorInvalidValue;
    }
}

template <int KS, int NT>
static hipError_t ngemm_launch_fused(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, NgFuse F,
                                     size_t prev_rows, hipStream_t s, bool* fused, uint64_t* d_progress, size_t progress_slots) {
    constexpr int WGB = 16 * kNgCompute * NT;
    const uint64_t batches = (L.len + P.k - 1) / P.k;
    const uint64_t chunks = (batches + WGB - 1) / WGB;
    F.n_gen = chunks * L.participants;
    const bool have_comb = F.prev && prev_rows > 0 && F.jobs > 0 && F.dimension > 0;
    // ---- the default: clerk WAVES inside the share-generation workgroups (see ng_clerk_wave), then the follow-up kernel for the rest
    const uint64_t slots = F.n_gen * (uint64_t)kNgClerkWaves;
    // (an even dimension: the clerk waves read and write the running sums of a lane's two columns as one 16-byte access)
    if (d_progress && have_comb && F.n_gen > 0 && F.n_gen <= 0x7FFFFFFFull && slots <= progress_slots && prev_rows < (1ull << 31) &&
        F.dimension % 2 == 0) {
        const uint64_t col_blocks = (((F.dimension + 1) / 2) + 63) / 64;
        if (col_blocks <= 0xFFFFFFFFull) {
            F.cw_col_blocks = (uint32_t)col_blocks;
            F.cw_items = col_blocks * F.jobs;
            F.cw_per_slot = (F.cw_items + slots - 1) / slots;
            F.cw_progress = d_progress;
            F.n_comb = 0; F.n_comb_wg = 0; F.n_early = 0; F.period = 1; F.items_per_wg = 0;
            const size_t lds = ngemm_lds_bytes<KS, NT>();
            auto kern = packed_gen_ngemm_kernel<KS, NT>;
            if (lds > 64 * 1024)
                if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
            *fused = true;
            note_kernel("packed_gen_ngemm_kernel<%d, %d>", KS, NT);
            kern<<<dim3((unsigned)F.n_gen), dim3(kNgThreads), lds, s>>>(L, mod, key, P, chunks ? chunks : 1, batches, F);
            if (hipError_t e = hipGetLastError()) return e;
            const uint64_t cap = 256ull * 32ull;                      // single-wave workgroups walking the slots
            ngemm_clerk_rest_kernel<<<dim3((unsigned)(slots < cap ? slots : cap)), dim3(64), 0, s>>>(F, slots);
            return hipGetLastError();
        }
    }
    F.col_blocks = (uint32_t)((((F.dimension + 1) / 2) + 255) / 256);
    // clerk-sum items of up to 512 rows (as in the other dual-role launches)
    uint64_t splits = have_comb ? (prev_rows + 511) / 512 : 1;
    if (splits > 64) splits = 64;
    F.rows_per_split = have_comb ? (prev_rows + splits - 1) / splits : 1;
    splits = have_comb ? (prev_rows + F.rows_per_split - 1) / F.rows_per_split : 1;
    F.splits = (uint32_t)splits;
    F.n_comb = have_comb ? (uint64_t)F.col_blocks * F.jobs * splits : 0;
    // Clerk workgroups: a pair of items each while that leaves at least two share-generation workgroups per clerk workgroup; a shape with
    // MORE clerk items than that (19682 clerk rows x 40 participants: 413 k items beside 1640 share-generation workgroups) gets
    // persistent clerk workgroups of items_per_wg items, so that the two roles still alternate in the grid.
    uint64_t per_wg = 2;
    const uint64_t most = F.n_gen == 0 ? (~0ull >> 2) : (F.n_gen / 2 ? F.n_gen / 2 : 1);    // (a clerk-only launch - the last of a run - keeps the pairs)
    if ((F.n_comb + 1) / 2 > most) per_wg = 2 * ((F.n_comb + 2 * most - 1) / (2 * most));
    if (per_wg > 0xFFFFFFFEull) return hipSuccess;                                // not fused: the caller issues the two launches
    F.items_per_wg = (uint32_t)per_wg;
    F.n_comb_wg = (F.n_comb + per_wg - 1) / per_wg;
    // Few share-generation workgroups per CU (a tile of n = 19682 is 6.4 rounds of them): the last round's workgroups would run beside
    // idle CUs.  A third of the clerk workgroups is kept back and FOLLOWS the last share-generation workgroup in the grid: short
    // items (a third of a share-generation workgroup's time each) that fill the tail.
    const uint32_t cus = ng_cu_count() ? ng_cu_count() : 256u;
    const uint64_t late = F.n_gen < 16ull * cus ? F.n_comb_wg / 3 : 0;
    F.n_early = F.n_comb_wg - late;
    // workgroup b runs on XCD b % 8: an odd period spreads the clerk positions over all of them
    F.period = F.n_early ? F.n_gen / F.n_early + 1 : 1;
    if (F.n_early && (F.period & 1) == 0) F.period = F.period > 2 ? F.period - 1 : 3;
    // (a smaller period than n_gen / n_early + 1 exhausts the early clerk positions before the share-generation items end: fine)
    uint64_t grid = F.n_gen + F.n_comb_wg;
    if (F.n_early && (F.n_early - 1) * F.period + 1 > F.n_gen + F.n_early) return hipSuccess;   // cannot happen (period <= n_gen / n_early + 1)
    if (grid == 0) { *fused = true; return hipSuccess; }
    if (grid > 0x7FFFFFFFull) return hipSuccess;                                  // not fused: the caller issues the two launches
    if (F.n_comb_wg == 0) { *fused = true; return ngemm_launch<KS, NT>(L, mod, key, P, s); }
    const size_t lds = ngemm_lds_bytes<KS, NT>();
    auto kern = packed_gen_ngemm_kernel<KS, NT>;
    if (lds > 64 * 1024)
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    *fused = true;
    note_kernel("packed_gen_ngemm_kernel<%d, %d>", KS, NT);
    kern<<<dim3((unsigned)grid), dim3(kNgThreads), lds, s>>>(L, mod, key, P, chunks ? chunks : 1, batches, F);
    return hipGetLastError();
}

// (participant, chunk) workgroups x clerk waves of a share-generation launch over L: the progress slots the caller provides
uint64_t ngemm_clerk_slots(const GenLayout& L, const NGemmPlan& P) {
    const int nt = P.ks == 1 ? 4 : P.ks == 8 ? 1 : 2;
    const uint64_t wgb = 16ull * kNgCompute * nt, batches = (L.len + P.k - 1) / P.k;
    return (batches + wgb - 1) / wgb * L.participants * (uint64_t)kNgClerkWaves;
}

hipError_t launch_fused_packed_ngemm(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, uint64_t* acc_lo,
                                     int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs, size_t dimension, hipStream_t s,
                                     bool* fused, uint64_t* d_progress, size_t progress_slots) {
    *fused = false;
    if (L.rand) return hipSuccess;
    if (L.participants == 0 || L.len == 0) return hipSuccess;       // nothing to generate: the plain clerk-sum kernel streams at the HBM rate
    const bool have_comb = d_prev && prev_rows > 0 && jobs > 0 && dimension > 0;
    // the clerk role reads the previous tile with 16-byte loads
    if (have_comb && !(((reinterpret_cast<uintptr_t>(d_prev) & 15u) == 0) && (L.out_stride_clerk % 2 == 0) && (L.out_stride_participant % 2 == 0)))
        return hipSuccess;
    NgFuse F{};
    F.acc_lo = acc_lo; F.acc_hi = acc_hi; F.prev = d_prev;
    F.job_stride = L.out_stride_clerk; F.row_stride = L.out_stride_participant;
    F.n_rows = prev_rows; F.dimension = dimension; F.jobs = (uint32_t)jobs;
    switch (P.ks) {
        case 1: return ngemm_launch_fused<1, 4>(L, mod, key, P, F, prev_rows, s, fused, d_progress, progress_slots);
        case 2: return ngemm_launch_fused<2, 2>(L, mod, key, P, F, prev_rows, s, fused, d_progress, progress_slots);
        case 4: return ngemm_launch_fused<4, 2>(L, mod, key, P, F, prev_rows, s, fused, d_progress, progress_slots);
        case 8: return ngemm_launch_fused<8, 1>(L, mod, key, P, F, prev_rows, s, fused, d_progress, progress_slots);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sda
