// RFC 4648 base64 of `Binary` payloads on the device: every `Encryption::Sodium(Binary)` travels
// as a base64 string inside JSON (protocol/src/helpers.rs:174-216, data_encoding::base64 - standard alphabet, '='
// padding, strict decoding).  One lane = 16 characters <-> 12 bytes; rows are independent messages.
//
// Both kernels are byte shuffles bound by HBM traffic (28 B per 12 payload bytes); nothing here is arithmetic on share
// values.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace sda {

static constexpr int kB64Threads = 256;

// character -> 6-bit value, or a value >= 64 for anything outside the alphabet ('=' included)
__device__ __forceinline__ uint32_t b64_value(uint32_t c) {
    const uint32_t up = c - 'A', lo = c - 'a', dg = c - '0';
    uint32_t v = 0xFF;
    v = up < 26u ? up : v;
    v = lo < 26u ? lo + 26u : v;
    v = dg < 10u ? dg + 52u : v;
    v = c == '+' ? 62u : v;
    v = c == '/' ? 63u : v;
    return v;
}
__device__ __forceinline__ uint32_t b64_char(uint32_t v) {      // 6-bit value -> character
    return v < 26u ? v + 'A' : v < 52u ? v + ('a' - 26u) : v < 62u ? v + ('0' - 52u) : v == 62u ? '+' : '/';
}

// text row r: d_text + (offsets ? offsets[r] : r * text_slot), lengths[r] characters.  Output row r: d_out + r * out_slot,
// its byte count -> out_bytes[r].  Malformed rows (length not a multiple of 4, a character outside the alphabet, '='
// anywhere but the last one or two positions, non-zero bits under the padding) set bit 8 of *status and row_status[r].
__global__ __launch_bounds__(kB64Threads) void base64_decode_rows_kernel(const uint8_t* __restrict__ text,
                                                                         const uint64_t* __restrict__ offsets,
                                                                         size_t text_slot, const uint64_t* __restrict__ lengths,
                                                                         uint8_t* __restrict__ out, size_t out_slot,
                                                                         uint64_t* __restrict__ out_bytes,
                                                                         uint32_t* __restrict__ status,
                                                                         uint32_t* __restrict__ row_status, size_t row0,
                                                                         uint64_t max_chars) {
    const size_t r = row0 + blockIdx.y;
    const uint64_t len = lengths[r];
    const uint64_t c0 = ((uint64_t)blockIdx.x * kB64Threads + threadIdx.x) * 16;      // first character of this lane
    if (len > max_chars) {
        // the length is network input (a job blob or a JSON scan): a row longer than the bound the buffers were sized for is
        // malformed - refused before a single byte of it is read or written (the grid only covers max_chars anyway)
        if (c0 == 0) { out_bytes[r] = 0; atomicOr(status, 8u); if (row_status) row_status[r] = 1; }
        return;
    }
    if (c0 == 0) {                                                                      // the row's byte count
        uint64_t nb = (len / 4) * 3;
        bool bad = (len & 3) != 0;
        if (!bad && len >= 4) {
            const uint8_t* t = text + (offsets ? offsets[r] : r * text_slot);
            if (t[len - 1] == '=') nb -= (t[len - 2] == '=') ? 2 : 1;
        }
        out_bytes[r] = bad ? 0 : nb;
        if (bad) { atomicOr(status, 8u); if (row_status) row_status[r] = 1; }
    }
    if (c0 >= len || (len & 3)) return;
    const uint8_t* src = text + (offsets ? offsets[r] : r * text_slot) + c0;
    // 16 characters through aligned dword loads and a byte funnel (the row may start anywhere inside a JSON document)
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(src - mis);
    const uint64_t avail = len - c0;                                                    // characters left in the row (>= 4)
    const uint32_t nq = avail >= 16 ? 4u : (uint32_t)(avail / 4);                       // quads of this lane
    uint32_t d[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = (uint32_t)(4 * i) < mis + 4 * nq ? w[i] : 0u;    // never read past the row's last dword
    uint32_t q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = mis ? __builtin_amdgcn_alignbyte(d[i + 1], d[i], mis) : d[i];
    uint8_t* dst = out + r * out_slot + (c0 / 4) * 3;
    bool bad = false;
    uint32_t o[3] = {0, 0, 0};                                                          // 12 output bytes, little-endian dwords
    uint32_t n_out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((uint32_t)i >= nq) break;
        const uint32_t ch0 = q[i] & 0xFF, ch1 = (q[i] >> 8) & 0xFF, ch2 = (q[i] >> 16) & 0xFF, ch3 = q[i] >> 24;
        const bool last = c0 + 4 * (uint64_t)i + 4 == len;                              // only the final quad may be padded
        uint32_t v0 = b64_value(ch0), v1 = b64_value(ch1), v2 = b64_value(ch2), v3 = b64_value(ch3);
        uint32_t bytes = 3;
        if (last && ch3 == '=') {
            v3 = 0; bytes = 2;
            if (ch2 == '=') { v2 = 0; bytes = 1; bad |= (v1 & 0xF) != 0; }              // strict: no stray bits
            else bad |= (v2 & 0x3) != 0;
        }
        bad |= (v0 | v1 | v2 | v3) > 63u;
        const uint32_t tri = (v0 << 18) | (v1 << 12) | (v2 << 6) | v3;                   // 24 bits, big-endian byte order
        const uint32_t b0 = tri >> 16, b1 = (tri >> 8) & 0xFF, b2 = tri & 0xFF;
        const uint32_t pos = 3 * i;
        const uint32_t bs[3] = {b0, b1, b2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t p = pos + j;
            if ((uint32_t)j < bytes) o[p >> 2] |= bs[j] << (8 * (p & 3));
        }
        n_out += bytes;
    }
    if (bad) { atomicOr(status, 8u); if (row_status) row_status[r] = 1; }
    if (n_out == 12) {
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);                                // dst is 4-byte aligned (slot and 12 * lane)
        d32[0] = o[0]; d32[1] = o[1]; d32[2] = o[2];
    } else {
        for (uint32_t p = 0; p < n_out; ++p) dst[p] = (uint8_t)(o[p >> 2] >> (8 * (p & 3)));
    }
}

// raw row r: d_in + r * in_slot (16-byte aligned), in_bytes[r] bytes -> text row r at d_text + r * text_slot, its
// length -> text_bytes[r]
__global__ __launch_bounds__(kB64Threads) void base64_encode_rows_kernel(const uint8_t* __restrict__ in, size_t in_slot,
                                                                         const uint64_t* __restrict__ in_bytes,
                                                                         uint8_t* __restrict__ text, size_t text_slot,
                                                                         uint64_t* __restrict__ text_bytes, size_t row0,
                                                                         uint64_t max_bytes) {
    const size_t r = row0 + blockIdx.y;
    const uint64_t n = in_bytes[r];
    const uint64_t b0 = ((uint64_t)blockIdx.x * kB64Threads + threadIdx.x) * 12;        // first byte of this lane
    if (n > max_bytes) {                      // longer than the slots were sized for: refused, nothing read or written
        if (b0 == 0) text_bytes[r] = 0;
        return;
    }
    if (b0 == 0) text_bytes[r] = (n + 2) / 3 * 4;
    if (b0 >= n) return;
    const uint8_t* src = in + r * in_slot + b0;
    const uint64_t left = n - b0;
    uint32_t w[3] = {0, 0, 0};
    if (left >= 12) {
        const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
        w[0] = s32[0]; w[1] = s32[1]; w[2] = s32[2];
    } else {
        for (uint32_t p = 0; p < (uint32_t)left; ++p) w[p >> 2] |= (uint32_t)src[p] << (8 * (p & 3));
    }
    uint32_t outw[4];
    const uint32_t nb = left >= 12 ? 12u : (uint32_t)left;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = 3 * i;
        const uint32_t x0 = (w[p >> 2] >> (8 * (p & 3))) & 0xFF, x1 = (w[(p + 1) >> 2] >> (8 * ((p + 1) & 3))) & 0xFF,
                       x2 = (w[(p + 2) >> 2] >> (8 * ((p + 2) & 3))) & 0xFF;
        const uint32_t tri = (x0 << 16) | (x1 << 8) | x2;
        const uint32_t have = nb > p ? (nb - p >= 3 ? 3u : nb - p) : 0u;
        uint32_t c0 = b64_char(tri >> 18), c1 = b64_char((tri >> 12) & 63), c2 = b64_char((tri >> 6) & 63), c3 = b64_char(tri & 63);
        if (have < 3) c3 = '=';
        if (have < 2) c2 = '=';
        outw[i] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    }
    uint8_t* dst = text + r * text_slot + (b0 / 3) * 4;                                  // 16-byte aligned: slot and 16 * lane
    const uint32_t quads = (nb + 2) / 3;
    if (quads == 4) {
        uint4 v; v.x = outw[0]; v.y = outw[1]; v.z = outw[2]; v.w = outw[3];
        *reinterpret_cast<uint4*>(dst) = v;
    } else {
        for (uint32_t i = 0; i < quads; ++i) reinterpret_cast<uint32_t*>(dst)[i] = outw[i];
    }
}

static inline uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

hipError_t launch_base64_decode_rows(const uint8_t* d_text, const uint64_t* d_offsets, size_t text_slot,
                                     const uint64_t* d_lengths, size_t rows, size_t max_chars, uint8_t* d_out, size_t out_slot,
                                     uint64_t* d_out_bytes, uint32_t* d_status, uint32_t* d_row_status, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    uint64_t chunks = cdiv(max_chars ? max_chars : 1, (uint64_t)kB64Threads * 16);
    if (chunks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    for (size_t r0 = 0; r0 < rows; r0 += 65535) {
        const unsigned nr = (unsigned)(rows - r0 < 65535 ? rows - r0 : 65535);
        base64_decode_rows_kernel<<<dim3((unsigned)chunks, nr), dim3(kB64Threads), 0, s>>>(d_text, d_offsets, text_slot, d_lengths, d_out,
                                                                                           out_slot, d_out_bytes, d_status, d_row_status, r0, max_chars);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_base64_encode_rows(const uint8_t* d_in, size_t in_slot, const uint64_t* d_in_bytes, size_t rows,
                                     size_t max_bytes, uint8_t* d_text, size_t text_slot, uint64_t* d_text_bytes, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    uint64_t chunks = cdiv(max_bytes ? max_bytes : 1, (uint64_t)kB64Threads * 12);
    if (chunks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    for (size_t r0 = 0; r0 < rows; r0 += 65535) {
        const unsigned nr = (unsigned)(rows - r0 < 65535 ? rows - r0 : 65535);
        base64_encode_rows_kernel<<<dim3((unsigned)chunks, nr), dim3(kB64Threads), 0, s>>>(d_in, in_slot, d_in_bytes, d_text, text_slot,
                                                                                           d_text_bytes, r0, max_bytes);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

}  // namespace sda
