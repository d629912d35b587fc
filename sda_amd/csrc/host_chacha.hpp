// Host-side ChaCha20 block (RFC 7539 state layout) for the per-call key derivation of sda-drbg-v1.
// Runs once per API call, never per element.
#pragma once
#include <stdint.h>
#include <string.h>

namespace sda {

inline uint32_t h_rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

inline void h_chacha20_block(const uint32_t key[8], uint32_t c12, uint32_t c13, uint32_t c14, uint32_t c15,
                             uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key[0], key[1], key[2], key[3],
                       key[4],      key[5],      key[6],      key[7],      c12,    c13,    c14,    c15};
    uint32_t x[16];
    memcpy(x, in, sizeof x);
#define SDA_HQR(a, b, c, d)                                   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = h_rotl32(x[d], 16);    \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = h_rotl32(x[b], 12);    \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = h_rotl32(x[d], 8);     \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = h_rotl32(x[b], 7);
    for (int r = 0; r < 10; ++r) {
        SDA_HQR(0, 4, 8, 12) SDA_HQR(1, 5, 9, 13) SDA_HQR(2, 6, 10, 14) SDA_HQR(3, 7, 11, 15)
        SDA_HQR(0, 5, 10, 15) SDA_HQR(1, 6, 11, 12) SDA_HQR(2, 7, 8, 13) SDA_HQR(3, 4, 9, 14)
    }
#undef SDA_HQR
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
    explicit_bzero(x, sizeof x);
    explicit_bzero(in, sizeof in);
}

// sda-drbg-v1 call key: words 0..7 of the ChaCha20 block keyed with the handle's master key, block counter =
// call index, nonce words = "sdak" "dfv1".  Every device-resident call of a handle runs under its own key, so a
// caller that repeats `first_participant` cannot repeat a keystream.
#define SDA_KDF_NONCE0 0x6b616473u   /* "sdak" */
#define SDA_KDF_NONCE1 0x31766664u   /* "dfv1" */
inline void h_drbg_call_key(const uint32_t master[8], uint64_t call_index, uint32_t out[8]) {
    uint32_t blk[16];
    h_chacha20_block(master, (uint32_t)call_index, (uint32_t)(call_index >> 32), SDA_KDF_NONCE0, SDA_KDF_NONCE1, blk);
    memcpy(out, blk, 32);
    explicit_bzero(blk, sizeof blk);
}

}  // namespace sda
