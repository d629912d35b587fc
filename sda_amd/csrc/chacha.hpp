// ChaCha block function for gfx950, two shapes:
//
//  * chacha_block_lane   - one lane computes a whole 16-word block (used where one lane consumes all
//                          8 u64 of a block in order: the rand-0.3-compatible mask expansion, and
//                          the rare retry stream of the DRBG).
//  * chacha_block_quad   - FOUR adjacent lanes (a DPP quad) compute one block cooperatively, lane c
//                          holding column c of the 4x4 state.  The diagonal rounds rotate rows 1..3
//                          across the quad with v_mov_b32_dpp quad_perm - no LDS, no ds_bpermute.
//                          Each lane ends up with 4 words = 2 u64 = exactly the randomness one lane
//                          needs for the two batches it shares.  This is the share-generation DRBG.
//
// State layout (same as RFC 7539 / rand 0.3): words 0..3 constants, 4..11 key, 12..15 counter/nonce.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace sda {

#define SDA_CHACHA_C0 0x61707865u
#define SDA_CHACHA_C1 0x3320646Eu
#define SDA_CHACHA_C2 0x79622D32u
#define SDA_CHACHA_C3 0x6B206574u

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) {
    return __builtin_rotateleft32(x, n);   // v_alignbit_b32
}

#define SDA_QR(a, b, c, d)                \
    a += b; d ^= a; d = rotl32(d, 16);    \
    c += d; b ^= c; b = rotl32(b, 12);    \
    a += b; d ^= a; d = rotl32(d, 8);     \
    c += d; b ^= c; b = rotl32(b, 7);

// ---- whole block in one lane -------------------------------------------------------------------
template <int ROUNDS>
__device__ __forceinline__ void chacha_block_lane(const uint32_t (&key)[8], uint32_t c12, uint32_t c13,
                                                  uint32_t c14, uint32_t c15, uint32_t (&out)[16]) {
    uint32_t x0 = SDA_CHACHA_C0, x1 = SDA_CHACHA_C1, x2 = SDA_CHACHA_C2, x3 = SDA_CHACHA_C3;
    uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3];
    uint32_t x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    uint32_t x12 = c12, x13 = c13, x14 = c14, x15 = c15;
#pragma unroll
    for (int r = 0; r < ROUNDS / 2; ++r) {
        SDA_QR(x0, x4, x8, x12) SDA_QR(x1, x5, x9, x13) SDA_QR(x2, x6, x10, x14) SDA_QR(x3, x7, x11, x15)
        SDA_QR(x0, x5, x10, x15) SDA_QR(x1, x6, x11, x12) SDA_QR(x2, x7, x8, x13) SDA_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + SDA_CHACHA_C0; out[1] = x1 + SDA_CHACHA_C1; out[2] = x2 + SDA_CHACHA_C2; out[3] = x3 + SDA_CHACHA_C3;
    out[4] = x4 + key[0]; out[5] = x5 + key[1]; out[6] = x6 + key[2]; out[7] = x7 + key[3];
    out[8] = x8 + key[4]; out[9] = x9 + key[5]; out[10] = x10 + key[6]; out[11] = x11 + key[7];
    out[12] = x12 + c12; out[13] = x13 + c13; out[14] = x14 + c14; out[15] = x15 + c15;
}

// ---- one block per DPP quad ------------------------------------------------------------------------
// quad_perm control: lane i reads lane ((ctrl >> 2i) & 3) of its quad
#define SDA_QP_ROT1 0x39   // [1,2,3,0] : lane i <- lane i+1
#define SDA_QP_ROT2 0x4E   // [2,3,0,1] : lane i <- lane i+2
#define SDA_QP_ROT3 0x93   // [3,0,1,2] : lane i <- lane i+3

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// Lane c (= lane id & 3) passes ITS column of the input state: (const_c, key[c], key[4+c], ctr_c)
// and receives output words {c, 4+c, 8+c, 12+c} in (o0, o1, o2, o3).
template <int ROUNDS>
__device__ __forceinline__ void chacha_block_quad(uint32_t in_a, uint32_t in_b, uint32_t in_c, uint32_t in_d,
                                                  uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    uint32_t a = in_a, b = in_b, c = in_c, d = in_d;
#pragma unroll
    for (int r = 0; r < ROUNDS / 2; ++r) {
        SDA_QR(a, b, c, d)                 // column round: every lane owns one column
        b = quad_perm<SDA_QP_ROT1>(b);     // diagonalise: row1 <<< 1, row2 <<< 2, row3 <<< 3
        c = quad_perm<SDA_QP_ROT2>(c);
        d = quad_perm<SDA_QP_ROT3>(d);
        SDA_QR(a, b, c, d)                 // diagonal round
        b = quad_perm<SDA_QP_ROT3>(b);     // undo
        c = quad_perm<SDA_QP_ROT2>(c);
        d = quad_perm<SDA_QP_ROT1>(d);
    }
    o0 = a + in_a; o1 = b + in_b; o2 = c + in_c; o3 = d + in_d;
}

}  // namespace sda
