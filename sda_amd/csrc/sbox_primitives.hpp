// Primitives of a libsodium sealed box (crypto_box_seal = X25519 + HSalsa20/XSalsa20 + Poly1305, nonce = BLAKE2b-24),
// written once for the device kernels (sealedbox_kernels.hip) and the host unit test (tests/cpp/sbox_primitives_test.cpp
// compiles this header with g++ and checks it against the published vectors).  Reference call sites:
// client/src/crypto/encryption/sodium.rs:43 (seal), :78 (open); algorithms: RFC 7748 (X25519), Bernstein's Salsa20 /
// XSalsa20 papers, RFC 8439 section 2.5 (Poly1305), RFC 7693 (BLAKE2b), libsodium's crypto_box_seal construction.
//
// Everything is straight-line integer code on 32-bit limbs with 64-bit products (v_mad_i64_i32 / v_mad_u64_u32 on
// gfx950): one lane = one 64-byte block / one 16-byte piece for the bulk; the device runs the X25519 ladder across a
// DPP quad (x25519_quad in sealedbox_kernels.hip, same field arithmetic), x25519() below is its one-lane form (host test).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBX_HD __host__ __device__ __forceinline__
#define SBX_HD_NOINLINE __host__ __device__ __noinline__
#else
#define SBX_HD inline
#define SBX_HD_NOINLINE inline
#endif

namespace sda {
namespace sbx {

// ================================================================================================================
// GF(2^255 - 19): ten signed limbs, radix 2^25.5 (limb i carries 26 bits for even i, 25 for odd i)
// ================================================================================================================
struct Fe { int32_t v[10]; };

SBX_HD int fe_bits(int i) { return (i & 1) ? 25 : 26; }

// 256-bit little-endian integer in 8 words, top bit ignored (RFC 7748 decodeUCoordinate)
SBX_HD void fe_from_words(Fe& h, const uint32_t w[8]) {
    int off = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int bits = fe_bits(i), wi = off >> 5, sh = off & 31;
        uint64_t win = w[wi];
        if (wi + 1 < 8) win |= (uint64_t)w[wi + 1] << 32;
        uint32_t limb = (uint32_t)(win >> sh) & ((1u << bits) - 1u);
        if (i == 9) limb &= (1u << 25) - 1u;         // bits 230..254; bit 255 is dropped
        h.v[i] = (int32_t)limb;
        off += bits;
    }
}

SBX_HD void fe_set(Fe& h, int32_t x) {
#pragma unroll
    for (int i = 0; i < 10; ++i) h.v[i] = 0;
    h.v[0] = x;
}
SBX_HD void fe_add(Fe& h, const Fe& f, const Fe& g) {
#pragma unroll
    for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] + g.v[i];
}
SBX_HD void fe_sub(Fe& h, const Fe& f, const Fe& g) {
#pragma unroll
    for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] - g.v[i];
}
// swap f and g iff b == 1
SBX_HD void fe_cswap(Fe& f, Fe& g, uint32_t b) {
    const int32_t m = -(int32_t)b;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int32_t x = (f.v[i] ^ g.v[i]) & m;
        f.v[i] ^= x; g.v[i] ^= x;
    }
}

// carry chain on 64-bit limb sums -> |h_i| <= 2^25 (even) / 2^24 (odd) (+ small)
SBX_HD void fe_carry(Fe& out, int64_t h[10]) {
    // two interleaved chains (0,1,2,3,4 | 5,6,7,8,9), then the wrap 9 -> 0 and one more step
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bits = fe_bits(i);
            const int64_t c = (h[i] + ((int64_t)1 << (bits - 1))) >> bits;
            h[i + 1] += c;
            h[i] -= c << bits;
        }
        const int64_t c9 = (h[9] + ((int64_t)1 << 24)) >> 25;
        h[0] += c9 * 19;
        h[9] -= c9 << 25;
    }
    const int64_t c0 = (h[0] + ((int64_t)1 << 25)) >> 26;
    h[1] += c0;
    h[0] -= c0 << 26;
#pragma unroll
    for (int i = 0; i < 10; ++i) out.v[i] = (int32_t)h[i];
}

// h = f * g.  Inputs |f_i|, |g_i| < 2^27 (sums / differences of two carried elements); 64-bit column sums.
SBX_HD void fe_mul(Fe& out, const Fe& f, const Fe& g) {
    int32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { g19[i] = 19 * g.v[i]; f2[i] = 2 * f.v[i]; }
    int64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const bool odd2 = (i & 1) && (j & 1);          // both limbs carry 25 bits: the product needs a factor 2
            const int32_t a = odd2 ? f2[i] : f.v[i];
            const int k = i + j;
            if (k < 10) h[k] += (int64_t)a * g.v[j];
            else h[k - 10] += (int64_t)a * g19[j];
        }
    }
    fe_carry(out, h);
}
// h = f * f: the 45 off-diagonal products once (doubled) + the 10 squares, instead of 100 products
SBX_HD void fe_sq(Fe& out, const Fe& f) {
    int32_t f2[10], f19[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { f2[i] = 2 * f.v[i]; f19[i] = 19 * f.v[i]; }
    int64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) h[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // diagonal f_i^2: an extra factor 2 when the limb carries 25 bits (odd i); 19 when it wraps (2i >= 10)
        {
            const int32_t a = (i & 1) ? f2[i] : f.v[i];
            const int k = 2 * i;
            if (k < 10) h[k] += (int64_t)a * f.v[i];
            else h[k - 10] += (int64_t)a * f19[i];
        }
#pragma unroll
        for (int j = i + 1; j < 10; ++j) {
            // 2 f_i f_j, doubled again when both limbs carry 25 bits: the factors go on f_i (2 or 4 f_i fits: |f_i| < 2^26.1)
            const bool odd2 = (i & 1) && (j & 1);
            const int32_t a = odd2 ? 2 * f2[i] : f2[i];
            const int k = i + j;
            if (k < 10) h[k] += (int64_t)a * f.v[j];
            else h[k - 10] += (int64_t)a * f19[j];
        }
    }
    fe_carry(out, h);
}

// h = f * 121665  (a24 of RFC 7748 section 5)
SBX_HD void fe_mul_a24(Fe& out, const Fe& f) {
    int64_t h[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) h[i] = (int64_t)f.v[i] * 121665;
    fe_carry(out, h);
}

// z^(p-2): the standard 254-squaring addition chain for 2^255 - 21
SBX_HD_NOINLINE void fe_invert(Fe& out, const Fe& z) {
    Fe t0, t1, t2, t3;
    fe_sq(t0, z);                                              // 2
    fe_sq(t1, t0); fe_sq(t1, t1);                              // 8
    fe_mul(t1, z, t1);                                         // 9
    fe_mul(t0, t0, t1);                                        // 11
    fe_sq(t2, t0);                                             // 22
    fe_mul(t1, t1, t2);                                        // 31 = 2^5 - 1
    fe_sq(t2, t1); for (int i = 1; i < 5; ++i) fe_sq(t2, t2);  // 2^10 - 2^5
    fe_mul(t1, t2, t1);                                        // 2^10 - 1
    fe_sq(t2, t1); for (int i = 1; i < 10; ++i) fe_sq(t2, t2); // 2^20 - 2^10
    fe_mul(t2, t2, t1);                                        // 2^20 - 1
    fe_sq(t3, t2); for (int i = 1; i < 20; ++i) fe_sq(t3, t3); // 2^40 - 2^20
    fe_mul(t2, t3, t2);                                        // 2^40 - 1
    fe_sq(t2, t2); for (int i = 1; i < 10; ++i) fe_sq(t2, t2); // 2^50 - 2^10
    fe_mul(t1, t2, t1);                                        // 2^50 - 1
    fe_sq(t2, t1); for (int i = 1; i < 50; ++i) fe_sq(t2, t2); // 2^100 - 2^50
    fe_mul(t2, t2, t1);                                        // 2^100 - 1
    fe_sq(t3, t2); for (int i = 1; i < 100; ++i) fe_sq(t3, t3);// 2^200 - 2^100
    fe_mul(t2, t3, t2);                                        // 2^200 - 1
    fe_sq(t2, t2); for (int i = 1; i < 50; ++i) fe_sq(t2, t2); // 2^250 - 2^50
    fe_mul(t1, t2, t1);                                        // 2^250 - 1
    fe_sq(t1, t1); for (int i = 1; i < 5; ++i) fe_sq(t1, t1);  // 2^255 - 2^5
    fe_mul(out, t1, t0);                                       // 2^255 - 21
}

// canonical little-endian bytes as 8 words
SBX_HD void fe_to_words(uint32_t w[8], const Fe& f) {
    int32_t h[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) h[i] = f.v[i];
    int32_t q = (19 * h[9] + (1 << 24)) >> 25;
#pragma unroll
    for (int i = 0; i < 10; ++i) q = (h[i] + q) >> fe_bits(i);
    h[0] += 19 * q;                                            // now in [0, 2^255) after the carries below
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bits = fe_bits(i);
        const int32_t c = h[i] >> bits;
        h[i + 1] += c;
        h[i] -= c << bits;
    }
    h[9] &= (1 << 25) - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = 0;
    int off = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int wi = off >> 5, sh = off & 31;
        const uint64_t v = (uint64_t)(uint32_t)h[i] << sh;
        w[wi] |= (uint32_t)v;
        if (wi + 1 < 8) w[wi + 1] |= (uint32_t)(v >> 32);
        off += fe_bits(i);
    }
}

// RFC 7748 section 5: X25519(k, u); k and u as 8 little-endian words; clamping applied here
SBX_HD_NOINLINE void x25519(uint32_t out[8], const uint32_t k_in[8], const uint32_t u[8]) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = k_in[i];
    k[0] &= 0xFFFFFFF8u;
    k[7] = (k[7] & 0x7FFFFFFFu) | 0x40000000u;
    Fe x1, x2, z2, x3, z3;
    fe_from_words(x1, u);
    fe_set(x2, 1); fe_set(z2, 0); x3 = x1; fe_set(z3, 1);
    uint32_t swap = 0;
    for (int t = 254; t >= 0; --t) {
        const uint32_t kt = (k[t >> 5] >> (t & 31)) & 1u;
        swap ^= kt;
        fe_cswap(x2, x3, swap);
        fe_cswap(z2, z3, swap);
        swap = kt;
        Fe A, AA, B, BB, E, C, D, DA, CB, t0;
        fe_add(A, x2, z2); fe_sq(AA, A);
        fe_sub(B, x2, z2); fe_sq(BB, B);
        fe_sub(E, AA, BB);
        fe_add(C, x3, z3); fe_sub(D, x3, z3);
        fe_mul(DA, D, A); fe_mul(CB, C, B);
        fe_add(t0, DA, CB); fe_sq(x3, t0);
        fe_sub(t0, DA, CB); fe_sq(t0, t0); fe_mul(z3, x1, t0);
        fe_mul(x2, AA, BB);
        fe_mul_a24(t0, E); fe_add(t0, AA, t0); fe_mul(z2, E, t0);
    }
    fe_cswap(x2, x3, swap);
    fe_cswap(z2, z3, swap);
    Fe zi;
    fe_invert(zi, z2);
    fe_mul(x2, x2, zi);
    fe_to_words(out, x2);
}

// ================================================================================================================
// Salsa20 core / HSalsa20
// ================================================================================================================
SBX_HD uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define SBX_QR(a, b, c, d)          \
    x[b] ^= rotl(x[a] + x[d], 7);   \
    x[c] ^= rotl(x[b] + x[a], 9);   \
    x[d] ^= rotl(x[c] + x[b], 13);  \
    x[a] ^= rotl(x[d] + x[c], 18);

SBX_HD void salsa20_rounds(uint32_t x[16]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        SBX_QR(0, 4, 8, 12) SBX_QR(5, 9, 13, 1) SBX_QR(10, 14, 2, 6) SBX_QR(15, 3, 7, 11)
        SBX_QR(0, 1, 2, 3) SBX_QR(5, 6, 7, 4) SBX_QR(10, 11, 8, 9) SBX_QR(15, 12, 13, 14)
    }
}

#define SBX_SIGMA0 0x61707865u
#define SBX_SIGMA1 0x3320646Eu
#define SBX_SIGMA2 0x79622D32u
#define SBX_SIGMA3 0x6B206574u

SBX_HD void salsa20_block(uint32_t out[16], const uint32_t key[8], uint32_t n0, uint32_t n1, uint64_t counter) {
    uint32_t in[16] = {SBX_SIGMA0, key[0], key[1], key[2], key[3], SBX_SIGMA1, n0, n1, (uint32_t)counter, (uint32_t)(counter >> 32),
                       SBX_SIGMA2, key[4], key[5], key[6], key[7], SBX_SIGMA3};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = in[i];
    salsa20_rounds(x);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

SBX_HD void hsalsa20(uint32_t out[8], const uint32_t key[8], const uint32_t in4[4]) {
    uint32_t x[16] = {SBX_SIGMA0, key[0], key[1], key[2], key[3], SBX_SIGMA1, in4[0], in4[1], in4[2], in4[3],
                      SBX_SIGMA2, key[4], key[5], key[6], key[7], SBX_SIGMA3};
    salsa20_rounds(x);
    out[0] = x[0]; out[1] = x[5]; out[2] = x[10]; out[3] = x[15];
    out[4] = x[6]; out[5] = x[7]; out[6] = x[8]; out[7] = x[9];
}

// ================================================================================================================
// BLAKE2b, unkeyed, one block of <= 128 bytes (the sealed-box nonce hashes 64 bytes: epk || pk)
// ================================================================================================================
SBX_HD uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

SBX_HD_NOINLINE void blake2b_one_block(uint64_t out[8], const uint64_t m[16], uint32_t msg_bytes, uint32_t out_bytes) {
    const uint64_t iv[8] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                            0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
    // sigma, 4 bits per entry, round r at bits [64 r, 64 r + 64)
    const uint8_t sigma[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t h[8], v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = iv[i];
    h[0] ^= 0x01010000ull ^ out_bytes;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = iv[i]; }
    v[12] ^= msg_bytes;
    v[14] = ~v[14];
#define SBX_G(a, b, c, d, x, y)                                             \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);               \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);               \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);               \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = sigma[r];
        SBX_G(0, 4, 8, 12, m[s[0]], m[s[1]]) SBX_G(1, 5, 9, 13, m[s[2]], m[s[3]])
        SBX_G(2, 6, 10, 14, m[s[4]], m[s[5]]) SBX_G(3, 7, 11, 15, m[s[6]], m[s[7]])
        SBX_G(0, 5, 10, 15, m[s[8]], m[s[9]]) SBX_G(1, 6, 11, 12, m[s[10]], m[s[11]])
        SBX_G(2, 7, 8, 13, m[s[12]], m[s[13]]) SBX_G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef SBX_G
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = h[i] ^ v[i] ^ v[i + 8];
}

// sealed-box nonce: the first 24 bytes of BLAKE2b-24(epk || pk) as 6 words
SBX_HD void seal_nonce(uint32_t nonce[6], const uint32_t epk[8], const uint32_t pk[8]) {
    uint64_t m[16], h[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = (uint64_t)epk[2 * i] | ((uint64_t)epk[2 * i + 1] << 32);
        m[4 + i] = (uint64_t)pk[2 * i] | ((uint64_t)pk[2 * i + 1] << 32);
    }
#pragma unroll
    for (int i = 8; i < 16; ++i) m[i] = 0;
    blake2b_one_block(h, m, 64, 24);
#pragma unroll
    for (int i = 0; i < 3; ++i) { nonce[2 * i] = (uint32_t)h[i]; nonce[2 * i + 1] = (uint32_t)(h[i] >> 32); }
}

// ================================================================================================================
// Poly1305 in GF(2^130 - 5): five 26-bit limbs
// ================================================================================================================
struct P26 { uint32_t v[5]; };

// 16 little-endian bytes (4 words) + the pad bit at 2^(8 * nbytes) (nbytes = 16 for a full piece)
SBX_HD void p26_from_piece(P26& h, const uint32_t w[4], uint32_t nbytes) {
    uint32_t t[5] = {w[0], w[1], w[2], w[3], 0};
    // mask bytes beyond nbytes, set the pad bit
    if (nbytes < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lo = 4 * i;
            if ((int)nbytes <= lo) t[i] = 0;
            else if ((int)nbytes < lo + 4) t[i] &= (1u << (8 * (nbytes - lo))) - 1u;
        }
        t[nbytes >> 2] |= 1u << (8 * (nbytes & 3));
    } else {
        t[4] = 1;
    }
    h.v[0] = t[0] & 0x3FFFFFF;
    h.v[1] = ((t[0] >> 26) | (t[1] << 6)) & 0x3FFFFFF;
    h.v[2] = ((t[1] >> 20) | (t[2] << 12)) & 0x3FFFFFF;
    h.v[3] = ((t[2] >> 14) | (t[3] << 18)) & 0x3FFFFFF;
    h.v[4] = (t[3] >> 8) | (t[4] << 24);
}

// r from the first 16 key bytes, clamped (RFC 8439 section 2.5)
SBX_HD void p26_clamped_r(P26& r, const uint32_t k[4]) {
    const uint32_t t0 = k[0] & 0x0FFFFFFFu, t1 = k[1] & 0x0FFFFFFCu, t2 = k[2] & 0x0FFFFFFCu, t3 = k[3] & 0x0FFFFFFCu;
    r.v[0] = t0 & 0x3FFFFFF;
    r.v[1] = ((t0 >> 26) | (t1 << 6)) & 0x3FFFFFF;
    r.v[2] = ((t1 >> 20) | (t2 << 12)) & 0x3FFFFFF;
    r.v[3] = ((t2 >> 14) | (t3 << 18)) & 0x3FFFFFF;
    r.v[4] = t3 >> 8;
}

// h = (a * b) mod 2^130 - 5, partially reduced: limbs < 2^26 (+ a small excess in limb 1).  Inputs: limbs < 2^27.
SBX_HD void p26_mul(P26& out, const P26& a, const P26& b) {
    const uint32_t s1 = b.v[1] * 5, s2 = b.v[2] * 5, s3 = b.v[3] * 5, s4 = b.v[4] * 5;
    uint64_t d0 = (uint64_t)a.v[0] * b.v[0] + (uint64_t)a.v[1] * s4 + (uint64_t)a.v[2] * s3 + (uint64_t)a.v[3] * s2 + (uint64_t)a.v[4] * s1;
    uint64_t d1 = (uint64_t)a.v[0] * b.v[1] + (uint64_t)a.v[1] * b.v[0] + (uint64_t)a.v[2] * s4 + (uint64_t)a.v[3] * s3 + (uint64_t)a.v[4] * s2;
    uint64_t d2 = (uint64_t)a.v[0] * b.v[2] + (uint64_t)a.v[1] * b.v[1] + (uint64_t)a.v[2] * b.v[0] + (uint64_t)a.v[3] * s4 + (uint64_t)a.v[4] * s3;
    uint64_t d3 = (uint64_t)a.v[0] * b.v[3] + (uint64_t)a.v[1] * b.v[2] + (uint64_t)a.v[2] * b.v[1] + (uint64_t)a.v[3] * b.v[0] + (uint64_t)a.v[4] * s4;
    uint64_t d4 = (uint64_t)a.v[0] * b.v[4] + (uint64_t)a.v[1] * b.v[3] + (uint64_t)a.v[2] * b.v[2] + (uint64_t)a.v[3] * b.v[1] + (uint64_t)a.v[4] * b.v[0];
    uint64_t c;
    c = d0 >> 26; d0 &= 0x3FFFFFF; d1 += c;
    c = d1 >> 26; d1 &= 0x3FFFFFF; d2 += c;
    c = d2 >> 26; d2 &= 0x3FFFFFF; d3 += c;
    c = d3 >> 26; d3 &= 0x3FFFFFF; d4 += c;
    c = d4 >> 26; d4 &= 0x3FFFFFF; d0 += c * 5;
    c = d0 >> 26; d0 &= 0x3FFFFFF; d1 += c;
    out.v[0] = (uint32_t)d0; out.v[1] = (uint32_t)d1; out.v[2] = (uint32_t)d2; out.v[3] = (uint32_t)d3; out.v[4] = (uint32_t)d4;
}
SBX_HD void p26_add(P26& h, const P26& a, const P26& b) {
#pragma unroll
    for (int i = 0; i < 5; ++i) h.v[i] = a.v[i] + b.v[i];
}
// limbs of any size below 2^31 -> partially reduced (< 2^26 each, small excess in limb 1)
SBX_HD void p26_carry(P26& h) {
    uint32_t c;
    c = h.v[0] >> 26; h.v[0] &= 0x3FFFFFF; h.v[1] += c;
    c = h.v[1] >> 26; h.v[1] &= 0x3FFFFFF; h.v[2] += c;
    c = h.v[2] >> 26; h.v[2] &= 0x3FFFFFF; h.v[3] += c;
    c = h.v[3] >> 26; h.v[3] &= 0x3FFFFFF; h.v[4] += c;
    c = h.v[4] >> 26; h.v[4] &= 0x3FFFFFF; h.v[0] += c * 5;
    c = h.v[0] >> 26; h.v[0] &= 0x3FFFFFF; h.v[1] += c;
}

// tag = (h mod 2^130 - 5) + s  mod 2^128, as 4 words
SBX_HD void p26_finish(uint32_t tag[4], const P26& hin, const uint32_t s[4]) {
    P26 h = hin;
    p26_carry(h);
    p26_carry(h);
    // g = h + 5 - 2^130; take g if it did not go negative
    uint32_t g[5], c;
    g[0] = h.v[0] + 5; c = g[0] >> 26; g[0] &= 0x3FFFFFF;
    g[1] = h.v[1] + c; c = g[1] >> 26; g[1] &= 0x3FFFFFF;
    g[2] = h.v[2] + c; c = g[2] >> 26; g[2] &= 0x3FFFFFF;
    g[3] = h.v[3] + c; c = g[3] >> 26; g[3] &= 0x3FFFFFF;
    g[4] = h.v[4] + c - (1u << 26);
    const uint32_t take_g = (g[4] >> 31) - 1u;                 // all ones iff g >= 0
#pragma unroll
    for (int i = 0; i < 5; ++i) h.v[i] = (h.v[i] & ~take_g) | (g[i] & take_g);
    const uint32_t h0 = h.v[0] | (h.v[1] << 26), h1 = (h.v[1] >> 6) | (h.v[2] << 20), h2 = (h.v[2] >> 12) | (h.v[3] << 14),
                   h3 = (h.v[3] >> 18) | (h.v[4] << 8);
    uint64_t f;
    f = (uint64_t)h0 + s[0]; tag[0] = (uint32_t)f;
    f = (uint64_t)h1 + s[1] + (f >> 32); tag[1] = (uint32_t)f;
    f = (uint64_t)h2 + s[2] + (f >> 32); tag[2] = (uint32_t)f;
    f = (uint64_t)h3 + s[3] + (f >> 32); tag[3] = (uint32_t)f;
}

}  // namespace sbx
}  // namespace sda
