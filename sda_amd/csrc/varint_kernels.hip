// Zig-zag LEB128 codec of share vectors on gfx950: the wire format either side
// of the path - client/src/crypto/encryption/sodium.rs:36-41 (encode: `share.encode_var`) and :83-89
// (decode: `Share::decode_var` until the reader is empty); integer-encoding 1.0 `VarInt for i64`.
//
// Variable-length coding is a scan problem:
//   encode: byte length per value (clz) -> workgroup sums -> exclusive scan -> bytes staged in LDS and
//           copied out with dword stores;
//   decode: a byte with the MSB clear terminates a value, so value index = number of terminators
//           before it: terminator counts per 4 KiB -> exclusive scan -> every terminator's owner lane
//           looks back <= 9 bytes in an LDS tile (16-byte halo) and assembles the value.
// Both are HBM streams (about 9 B of wire + 8 B of value per 62-bit share).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_internal.hpp"
#include "kernels.hpp"
#include "modarith.hpp"

namespace sda {

static constexpr int kVT = 256;          // threads per workgroup
static constexpr int kVals = 8;          // encode: values per lane  (2048 per workgroup)
static constexpr int kBytes = 16;        // decode: bytes per lane   (4096 per workgroup)

__device__ __forceinline__ uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
__device__ __forceinline__ uint32_t varint_len(uint64_t zz) {
    const uint32_t x = (64u - (uint32_t)__clzll(zz | 1ull)) + 6u;     // bits + 6, in 7..70
    return (x * 37u) >> 8;                                            // x / 7 for x <= 70
}

// workgroup exclusive scan of one u32 per lane; returns the lane's prefix, *total = workgroup sum
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* lds_waves, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) lds_waves[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kVT / 64; ++w) {
        const uint32_t t = lds_waves[w];
        if (w < wave) off += t;
        tot += t;
    }
    *total = tot;
    __syncthreads();
    return off + incl - v;
}

// ---- encode ---------------------------------------------------------------------------------------
// Only the workgroup total matters here, so the 2048 values of a block are dealt to the lanes in coalesced pairs
// (pair q of the block -> lane q % 256, one 16-byte load when the pair lies inside a row and is aligned) instead of
// the 8 consecutive values per lane the write kernel needs.
__global__ __launch_bounds__(kVT) void varint_len_kernel(VarintRows R, uint32_t* __restrict__ block_bytes) {
    __shared__ uint32_t waves[kVT / 64];
    const uint64_t N = (uint64_t)R.rows * R.len;
    const uint64_t base = (uint64_t)blockIdx.x * kVT * kVals;
    const bool vec_ok = (((uintptr_t)R.values) & 15u) == 0 && (R.row_stride & 1u) == 0;
    uint32_t sum = 0;
    uint64_t g = base + 2 * (uint64_t)threadIdx.x;
    if (g < N) {
        uint64_t r = g / R.len, i = g - r * R.len;
#pragma unroll
        for (int u = 0; u < kVals / 2; ++u) {
            if (g < N) {
                const int64_t* p = R.values + r * R.row_stride + i;
                if (vec_ok && i + 1 < R.len && (i & 1u) == 0) {
                    typedef long long ll2v __attribute__((ext_vector_type(2)));
                    const ll2v v = __builtin_nontemporal_load(reinterpret_cast<const ll2v*>(p));
                    sum += varint_len(zigzag(v.x)) + varint_len(zigzag(v.y));
                } else {
                    sum += varint_len(zigzag(p[0]));
                    if (g + 1 < N) sum += varint_len(zigzag(i + 1 < R.len ? p[1] : R.values[(r + 1) * R.row_stride]));
                }
            }
            g += 2 * kVT;
            i += 2 * kVT;
            if (i >= R.len) { const uint64_t q = i / R.len; r += q; i -= q * R.len; }
        }
    }
    uint32_t total;
    (void)block_exscan(sum, waves, &total);
    if (threadIdx.x == 0) block_bytes[blockIdx.x] = total;
}

// ---- exclusive scan of u32 -> u64, three small kernels: scan 1024-entry chunks in parallel, scan the
// chunk totals with one workgroup, add the chunk offsets back ------------------------------------------------
__device__ __forceinline__ uint64_t block1024_exscan(uint64_t v, uint64_t* wave_sum, uint64_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint64_t t = wave_sum[w];
        if (w < wave) off += t;
        tot += t;
    }
    *total = tot;
    __syncthreads();
    return off + incl - v;
}

__global__ __launch_bounds__(1024) void scan_chunks_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out,
                                                           uint64_t* __restrict__ chunk_tot, size_t n) {
    __shared__ uint64_t wave_sum[16];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint64_t total;
    const uint64_t ex = block1024_exscan(i < n ? in[i] : 0, wave_sum, &total);
    if (i < n) out[i] = ex;
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}

// one workgroup: in-place exclusive scan of the chunk totals, *total = grand total
__global__ __launch_bounds__(1024) void scan_totals_kernel(uint64_t* __restrict__ tot, size_t m, uint64_t* __restrict__ total) {
    __shared__ uint64_t wave_sum[16];
    uint64_t carry = 0;
    for (size_t base = 0; base < m; base += 1024) {
        const size_t i = base + threadIdx.x;
        uint64_t t;
        const uint64_t ex = block1024_exscan(i < m ? tot[i] : 0, wave_sum, &t);
        if (i < m) tot[i] = carry + ex;
        carry += t;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(1024) void scan_add_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ chunk_off,
                                                        size_t n) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += chunk_off[blockIdx.x];
}

__global__ __launch_bounds__(kVT) void varint_write_kernel(VarintRows R, const uint64_t* __restrict__ block_off,
                                                           uint8_t* __restrict__ out, uint64_t* __restrict__ row_offsets) {
    __shared__ uint32_t waves[kVT / 64];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kVT * kVals * 10 + 16];
    const uint64_t N = (uint64_t)R.rows * R.len;
    const uint64_t g0 = ((uint64_t)blockIdx.x * kVT + threadIdx.x) * kVals;
    uint64_t zz[kVals];
    uint32_t ln[kVals];
    uint32_t sum = 0;
    uint64_t r0 = 0, i0 = 0;
    if (g0 < N) { r0 = g0 / R.len; i0 = g0 - r0 * R.len; }
    {
        uint64_t r = r0, i = i0;
#pragma unroll
        for (int k = 0; k < kVals; ++k) {
            zz[k] = 0; ln[k] = 0;
            if (g0 + k < N) {
                zz[k] = zigzag(R.values[r * R.row_stride + i]);
                ln[k] = varint_len(zz[k]);
                sum += ln[k];
            }
            if (++i == R.len) { i = 0; ++r; }
        }
    }
    uint32_t total;
    uint32_t pos = block_exscan(sum, waves, &total);
    const uint64_t boff = block_off[blockIdx.x];
    {
        uint64_t r = r0, i = i0;
#pragma unroll
        for (int k = 0; k < kVals; ++k) {
            if (g0 + k < N) {
                if (i == 0 && row_offsets) row_offsets[r] = boff + pos;     // this value opens row r
                uint64_t n = zz[k];
                for (uint32_t b = 0; b + 1 < ln[k]; ++b) { stage[pos++] = (uint8_t)(0x80u | (n & 0x7Fu)); n >>= 7; }
                stage[pos++] = (uint8_t)n;
            }
            if (++i == R.len) { i = 0; ++r; }
        }
    }
    __syncthreads();
    // copy out: bytes up to the first 4-byte boundary of the destination, then dwords, then the tail
    uint8_t* dst = out + boff;
    const uint32_t head = (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u);
    const uint32_t h = head < total ? head : total;
    if (threadIdx.x < h) dst[threadIdx.x] = stage[threadIdx.x];
    const uint32_t n_dw = (total - h) >> 2;
    const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);
    uint32_t* dst32 = reinterpret_cast<uint32_t*>(dst + h);
    for (uint32_t d = threadIdx.x; d < n_dw; d += kVT) {
        const uint32_t byte = h + 4u * d;                  // source byte offset in stage
        const uint32_t lo = stage32[byte >> 2], hi = stage32[(byte >> 2) + 1];
        dst32[d] = __builtin_amdgcn_alignbyte(hi, lo, byte & 3u);
    }
    const uint32_t done = h + 4u * n_dw;
    if (threadIdx.x < total - done) dst[done + threadIdx.x] = stage[done + threadIdx.x];
}

// ---- decode -----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t count_terminators16(const uint32_t (&w)[4]) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += __builtin_popcount(~w[k] & 0x80808080u);
    return c;
}

__device__ __forceinline__ void load16(const uint8_t* __restrict__ bytes, uint64_t n_bytes, uint64_t p, uint32_t (&w)[4]) {
    if (p + 16 <= n_bytes && ((uintptr_t)(bytes + p) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(bytes + p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {                                            // ragged tail / unaligned base: bytes beyond the end read as
#pragma unroll                                          // continuation bytes (0x80) so they are never terminators
        for (int k = 0; k < 4; ++k) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint64_t q = p + 4 * k + b;
                x |= (uint32_t)(q < n_bytes ? bytes[q] : 0x80u) << (8 * b);
            }
            w[k] = x;
        }
    }
}

__global__ __launch_bounds__(kVT) void varint_count_kernel(const uint8_t* __restrict__ bytes, uint64_t n_bytes,
                                                           uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t waves[kVT / 64];
    const uint64_t p = ((uint64_t)blockIdx.x * kVT + threadIdx.x) * kBytes;
    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    if (p < n_bytes) load16(bytes, n_bytes, p, w);
    uint32_t total;
    (void)block_exscan(count_terminators16(w), waves, &total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// status bits
#define SDA_VARINT_MALFORMED 1u      // more than 10 bytes without a terminator
#define SDA_VARINT_ROW_COUNT 2u      // a row does not hold exactly `len` values
#define SDA_VARINT_UNTERMINATED 4u   // a row (or the stream) ends inside a value

__global__ __launch_bounds__(kVT) void varint_decode_kernel(const uint8_t* __restrict__ bytes, uint64_t n_bytes,
                                                            const uint64_t* __restrict__ block_val_off, uint64_t rows,
                                                            uint64_t len, uint64_t row_stride,
                                                            int64_t* __restrict__ out, uint32_t* __restrict__ status) {
    __shared__ uint32_t waves[kVT / 64];
    __shared__ __attribute__((aligned(16))) uint8_t tile[16 + kVT * kBytes + 16];   // halo | span | read slack
    const uint64_t block_base = (uint64_t)blockIdx.x * kVT * kBytes;
    const uint64_t p = block_base + (uint64_t)threadIdx.x * kBytes;
    uint32_t w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    if (p < n_bytes) load16(bytes, n_bytes, p, w);
    uint32_t* tile32 = reinterpret_cast<uint32_t*>(tile);
#pragma unroll
    for (int k = 0; k < 4; ++k) tile32[4 + threadIdx.x * 4 + k] = w[k];
    if (threadIdx.x < 16) {                                 // halo: the 16 bytes before this workgroup's span
        const uint64_t q = block_base + threadIdx.x;
        tile[threadIdx.x] = q >= 16 ? bytes[q - 16] : 0u;   // before the stream: a terminator stops the look-back
    }
    uint32_t total;
    uint32_t idx = block_exscan(count_terminators16(w), waves, &total);   // also orders the tile writes
    uint64_t g = block_val_off[blockIdx.x] + idx;
    const uint64_t N = rows * len;
    uint64_t r = 0, i = 0;
    bool have_ri = false;
    // continuation-bit map of this lane's 16 bytes (bit k = MSB of byte k) and of the 16 bytes before it
    uint32_t own = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = w[k] & 0x80808080u;                      // bits 7,15,23,31 -> 0..3
        own |= (((m >> 7) | (m >> 14) | (m >> 21) | (m >> 28)) & 0xFu) << (4 * k);
    }
    uint32_t prev = 0;
    {
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile) + threadIdx.x * 4;   // the 16 bytes before
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t m = t32[k] & 0x80808080u;
            prev |= (((m >> 7) | (m >> 14) | (m >> 21) | (m >> 28)) & 0xFu) << (4 * k);
        }
        // (the halo before the start of the stream was filled with terminators, so prev == 0 there)
    }
    const uint32_t cont = (own << 16) | prev;                       // 32-byte window, bit q = byte q continues
    uint32_t terms = ~own & 0xFFFFu;
    if (p + 16 > n_bytes) terms &= p < n_bytes ? (1u << (uint32_t)(n_bytes - p)) - 1u : 0u;
    while (terms) {
        const int k = __builtin_ctz(terms);
        terms &= terms - 1;
        const int e = 16 + k;                                       // window position of the terminator
        // previous non-continuation byte below e: the value starts right after it
        const uint32_t below = ~cont & ((1u << e) - 1u);
        const int prev_term = below ? 31 - __builtin_clz(below) : -1;
        int nb = e - prev_term;                                     // bytes of this value
        if (nb > 10) { atomicOr(status, SDA_VARINT_MALFORMED); nb = 10; }
        const int start = 16 + (int)threadIdx.x * kBytes + k - (nb - 1);   // tile coordinate of the first byte
        // 12 bytes from the tile at an arbitrary byte offset
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile) + (start >> 2);
        const uint32_t sh = (uint32_t)start & 3u;
        const uint32_t d0 = __builtin_amdgcn_alignbyte(t32[1], t32[0], sh);
        const uint32_t d1 = __builtin_amdgcn_alignbyte(t32[2], t32[1], sh);
        const uint32_t d2 = __builtin_amdgcn_alignbyte(t32[3], t32[2], sh);
        uint64_t x = ((uint64_t)d1 << 32) | d0;                     // bytes 0..7 of the value
        if (nb < 8) x &= (1ull << (8 * nb)) - 1ull;
        x &= 0x7F7F7F7F7F7F7F7Full;                                 // drop the continuation bits, then squeeze
        x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
        x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
        x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
        if (nb > 8) x |= (uint64_t)(d2 & 0x7Fu) << 56;
        if (nb > 9) x |= (uint64_t)((d2 >> 8) & 0x7Fu) << 63;
        const int64_t v = (int64_t)((x >> 1) ^ (uint64_t)(-(int64_t)(x & 1)));
        if (g < N) {
            if (!have_ri) { r = g / len; i = g - r * len; have_ri = true; }
            out[r * row_stride + i] = v;
            if (++i == len) { i = 0; ++r; }
        }
        ++g;
    }
}

// terminators in [block start, x) counted by a whole wave: 64 lanes x 16 B per step
__device__ __forceinline__ uint64_t wave_prefix(const uint8_t* __restrict__ bytes, uint64_t n_bytes,
                                                const uint64_t* __restrict__ block_val_off, uint64_t x) {
    const uint64_t blk = x / (kVT * kBytes);
    const uint64_t base = blk * (kVT * kBytes);
    const int lane = threadIdx.x & 63;
    uint32_t c = 0;
    for (uint64_t q = base + (uint64_t)lane * 16; q < x; q += 64 * 16) {
        uint32_t w[4];
        load16(bytes, n_bytes, q, w);
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t m = ~w[k] & 0x80808080u;
            t |= (((m >> 7) | (m >> 14) | (m >> 21) | (m >> 28)) & 0xFu) << (4 * k);
        }
        if (q + 16 > x) t &= (1u << (uint32_t)(x - q)) - 1u;
        c += __builtin_popcount(t);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    return block_val_off[blk] + c;
}

// one WAVE per row: the row must end on a terminator and hold exactly `len` values
__global__ __launch_bounds__(kVT) void varint_rowcheck_kernel(const uint8_t* __restrict__ bytes, uint64_t n_bytes,
                                                              const uint64_t* __restrict__ offsets, uint64_t rows,
                                                              uint64_t len, const uint64_t* __restrict__ block_val_off,
                                                              uint32_t* __restrict__ status) {
    const uint64_t r = (uint64_t)blockIdx.x * (kVT / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bool lead = (threadIdx.x & 63) == 0;
    const uint64_t a = offsets ? offsets[r] : 0, b = offsets ? offsets[r + 1] : n_bytes;
    if (b < a || b > n_bytes) { if (lead) atomicOr(status, SDA_VARINT_ROW_COUNT); return; }
    if (len == 0) { if (a != b && lead) atomicOr(status, SDA_VARINT_ROW_COUNT); return; }
    if (a == b || (bytes[b - 1] & 0x80u)) { if (lead) atomicOr(status, SDA_VARINT_UNTERMINATED); return; }
    const uint64_t cnt = wave_prefix(bytes, n_bytes, block_val_off, b) - wave_prefix(bytes, n_bytes, block_val_off, a);
    if (cnt != len && lead) atomicOr(status, SDA_VARINT_ROW_COUNT);
}

// ---- single-pass row streaming ----------------------------------------------------------------------
// Every encoded vector (row) is its own message with a known byte range, so the only sequential dependence is
// inside a row.  One WAVE streams one row in fixed 1 KiB chunks (64 lanes x 16 B) laid on the 16-byte grid below
// the row start: chunk addresses do not depend on the content, so the next chunks are prefetched into registers
// while the current one is decoded; what is carried from chunk to chunk is the column index, the continuation
// bitmap of the last 16 bytes and those 16 bytes themselves (the halo a value crossing the chunk boundary needs).
// The bytes are read ONCE (the three-pass form reads them twice and needs the block scan in between).
static constexpr int kStreamWaves = 4;                 // rows per workgroup
static constexpr int kStreamDepth = 4;                 // chunks in flight per wave
static constexpr int kStreamTile = 16 + 64 * 16 + 16;  // halo | chunk | read slack, per wave


// bit k = MSB of byte k.  (m >> 7) has the four flags at bits 0, 8, 16, 24; the multiplier moves them to bits
// 28..31 (partial products below bit 28 never collide, so nothing carries in).
__device__ __forceinline__ uint32_t cont_bits16(const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) r |= ((((w[k] & 0x80808080u) >> 7) * 0x10204080u) >> 28) << (4 * k);
    return r;
}

// wave64 inclusive scan with DPP row shifts / broadcasts (no LDS crossbar, no waits)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x, uint32_t src) {
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    uint32_t x = dpp_add<0x111, 0xf, 0xf>(v, v);      // row_shr:1
    x = dpp_add<0x112, 0xf, 0xf>(x, v);               // row_shr:2
    x = dpp_add<0x113, 0xf, 0xf>(x, v);               // row_shr:3  -> lanes hold the sum of up to 4 neighbours
    x = dpp_add<0x114, 0xf, 0xe>(x, x);               // row_shr:4, banks 1-3
    x = dpp_add<0x118, 0xf, 0xc>(x, x);               // row_shr:8, banks 2-3 -> inclusive scan inside each row of 16
    x = dpp_add<0x142, 0xa, 0xf>(x, x);               // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc, 0xf>(x, x);               // row_bcast:31 into rows 2 and 3
    return x;
}

// 16 bytes at bytes + rel (rel may reach 15 bytes below the buffer: same 16-byte granule as its first byte).
// Addressed from the kernel argument so that it stays a GLOBAL load: a flat load would also count as an LDS
// operation and every LDS wait would then drain the prefetch queue.
__device__ __forceinline__ uint4 stream_load(const uint8_t* __restrict__ bytes, int64_t rel, int64_t end_rel) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (rel < end_rel) {                                           // aligned, and the granule holds a byte of the row
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bytes + rel));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
}

// One value out of the LDS tile: `start` = tile coordinate of its first byte, nb = 1..10 bytes.
__device__ __forceinline__ int64_t tile_value(const uint32_t* tile32, int start, int nb) {
    const uint32_t* t32 = tile32 + (start >> 2);
    const uint32_t sh = (uint32_t)start & 3u;
    uint32_t d0 = __builtin_amdgcn_alignbyte(t32[1], t32[0], sh);       // bytes 0..3
    uint32_t d1 = __builtin_amdgcn_alignbyte(t32[2], t32[1], sh);       // bytes 4..7
    const uint32_t d2 = __builtin_amdgcn_alignbyte(t32[3], t32[2], sh); // bytes 8..9 (+2 foreign)
    // squeeze 4 x 7 bits per dword (the MSBs are dropped by the masks)
    d0 = (d0 & 0x007F007Fu) | ((d0 >> 1) & 0x3F803F80u);
    d1 = (d1 & 0x007F007Fu) | ((d1 >> 1) & 0x3F803F80u);
    d0 = (d0 & 0x00003FFFu) | ((d0 >> 2) & 0x0FFFC000u);
    d1 = (d1 & 0x00003FFFu) | ((d1 >> 2) & 0x0FFFC000u);
    uint64_t x = (uint64_t)d0 | ((uint64_t)d1 << 28);                   // 56 bits of bytes 0..7
    const int keep = 7 * (nb < 8 ? nb : 8);                             // bits that belong to this value
    x = (x << (64 - keep)) >> (64 - keep);
    const uint32_t top = (nb > 8 ? (d2 & 0x7Fu) : 0u) | (nb > 9 ? (d2 & 0x100u) >> 1 : 0u);   // byte 8, bit 0 of byte 9
    x |= (uint64_t)top << 56;
    return (int64_t)((x >> 1) ^ (uint64_t)(-(int64_t)(x & 1)));         // zig-zag
}

// A row being streamed by one wave.  next_group() decodes up to kStreamDepth chunks and hands every value to
// sink(column, value); the state between calls is the column index, the continuation bitmap of the last 16 bytes,
// the halo in the wave's LDS tile and the prefetched chunks.
struct RowStream {
    const uint8_t* bytes;
    int64_t rel0, end_rel;            // this lane's byte offset in chunk 0; end of the row
    int lo0;                          // bytes of this lane that precede the row in chunk 0 (0..16)
    uint64_t n_chunks, j0, len, col;
    uint32_t carry_prev, flags;
    uint32_t* tile32;
    uint4 w[kStreamDepth];

    __device__ __forceinline__ void open(const uint8_t* __restrict__ bytes_, uint64_t a, uint64_t b, uint64_t len_, uint8_t* tile) {
        const int lane = threadIdx.x & 63;
        bytes = bytes_;
        const uintptr_t start_addr = (uintptr_t)bytes + a;
        const uintptr_t base0 = start_addr & ~(uintptr_t)15;
        n_chunks = ((uintptr_t)bytes + b - base0 + 1023) / 1024;
        rel0 = (int64_t)(base0 - (uintptr_t)bytes) + lane * 16;
        end_rel = (int64_t)b;
        const int64_t before = (int64_t)a - rel0;
        lo0 = before <= 0 ? 0 : (before >= 16 ? 16 : (int)before);
        j0 = 0; len = len_; col = 0; carry_prev = 0; flags = 0;   // before the row every byte "terminates"
        tile32 = reinterpret_cast<uint32_t*>(tile);
#pragma unroll
        for (int d = 0; d < kStreamDepth; ++d) w[d] = stream_load(bytes, rel0 + (int64_t)d * 1024, end_rel);
    }
    __device__ __forceinline__ bool done() const { return j0 >= n_chunks; }

    template <class Sink>
    __device__ __forceinline__ void next_group(Sink&& sink) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int d = 0; d < kStreamDepth; ++d) {
            const uint64_t j = j0 + d;
            if (j >= n_chunks) continue;                          // wave-uniform
            uint4 cur = w[d];
            // Refill the slot only after `cur` has landed: the variable number of stores in the decode loop makes
            // the compiler wait for EVERY outstanding memory operation (vmcnt(0)) at the first use of `cur`; the
            // empty asm is that first use, so the wait sits before the new load, which then has a whole chunk's
            // work to arrive while the other kStreamDepth - 1 slots have long been in flight.
            asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w));
            __builtin_amdgcn_sched_barrier(0);
            w[d] = stream_load(bytes, rel0 + (int64_t)(j + kStreamDepth) * 1024, end_rel);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t own = cont_bits16(cur), terms = ~own & 0xFFFFu;
            if (j == 0 || j + 1 >= n_chunks) {                    // wave-uniform: only the edge chunks hold foreign bytes
                const int64_t q = rel0 + (int64_t)j * 1024;
                const int lo = j == 0 ? lo0 : 0;
                const int hi = q >= end_rel ? 0 : (end_rel - q >= 16 ? 16 : (int)(end_rel - q));
                const uint32_t below_lo = (1u << lo) - 1u;
                own &= ~below_lo;                                 // bytes before the row stop the look-back
                terms = ~own & ((1u << hi) - 1u) & ~below_lo & 0xFFFFu;
            }
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_prev, (int)own, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            carry_prev = __builtin_amdgcn_readlane(own, 63);
            const uint32_t cnt = __builtin_popcount(terms);
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            uint64_t g = col + (incl - cnt);
            // stage the chunk behind the halo (LDS operations of one wave execute in order)
            tile32[4 + lane * 4 + 0] = cur.x; tile32[4 + lane * 4 + 1] = cur.y;
            tile32[4 + lane * 4 + 2] = cur.z; tile32[4 + lane * 4 + 3] = cur.w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t cont = (own << 16) | prev;
            uint32_t t = terms;
            while (t) {
                const int k = __builtin_ctz(t);
                t &= t - 1;
                const int e = 16 + k;
                const uint32_t below = ~cont & ((1u << e) - 1u);
                const int prev_term = 31 - __builtin_clz(below | 1u);   // none in the window: nb >= 16, refused either way
                int nb = e - prev_term;
                if (nb > 10) { flags |= SDA_VARINT_MALFORMED; nb = 10; }
                const int64_t v = tile_value(tile32, 16 + lane * 16 + k - (nb - 1), nb);
                if (g < len) sink(g, v);
                ++g;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 63) { tile32[0] = cur.x; tile32[1] = cur.y; tile32[2] = cur.z; tile32[3] = cur.w; }   // next halo
            col += total;
        }
        j0 += kStreamDepth;
    }
    // row verdict once the stream is exhausted
    __device__ __forceinline__ void close(uint32_t* __restrict__ status) {
        if (col != len) flags |= SDA_VARINT_ROW_COUNT;
        if (flags) atomicOr(status, flags);
    }
};

struct StoreSink {
    int64_t* row;
    __device__ __forceinline__ void operator()(uint64_t c, int64_t v) const { row[c] = v; }
};

// row checks shared by the streaming kernels; true = stream the row
__device__ __forceinline__ bool stream_row_range(const uint8_t* __restrict__ bytes, uint64_t n_bytes,
                                                 const RowRanges& rr, uint64_t r, uint64_t len,
                                                 uint32_t* __restrict__ status, uint64_t& a, uint64_t& b) {
    const bool lead = (threadIdx.x & 63) == 0;
    if (rr.lengths) {
        a = r * rr.slot; b = a + rr.lengths[r];
        if (rr.lengths[r] > rr.slot) { if (lead) atomicOr(status, SDA_VARINT_ROW_COUNT); return false; }   // a row never leaves its slot
    } else { a = rr.offsets ? rr.offsets[r] : 0; b = rr.offsets ? rr.offsets[r + 1] : n_bytes; }
    if (b < a || b > n_bytes) { if (lead) atomicOr(status, SDA_VARINT_ROW_COUNT); return false; }
    if (len == 0) { if (a != b && lead) atomicOr(status, SDA_VARINT_ROW_COUNT); return false; }
    if (a == b || (bytes[b - 1] & 0x80u)) { if (lead) atomicOr(status, SDA_VARINT_UNTERMINATED); return false; }
    return true;
}

__global__ __launch_bounds__(kStreamWaves * 64) void varint_stream_decode_kernel(const uint8_t* __restrict__ bytes,
                                                                                 uint64_t n_bytes, RowRanges offsets,
                                                                                 uint64_t rows, uint64_t len,
                                                                                 uint64_t row_stride,
                                                                                 int64_t* __restrict__ out,
                                                                                 uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[kStreamWaves][kStreamTile];
    const int wave = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * kStreamWaves + wave;
    if (r >= rows) return;
    uint64_t a, b;
    if (!stream_row_range(bytes, n_bytes, offsets, r, len, status, a, b)) return;
    RowStream rs;
    rs.open(bytes, a, b, len, tiles[wave]);
    const StoreSink sink{out + r * row_stride};
    while (!rs.done()) rs.next_group(sink);
    rs.close(status);
}

// ---- single-pass encode into slots ------------------------------------------------------------------------------
// The mirror image: one wave encodes one row, 128 values (2 per lane, one 16-byte load) per step, four steps
// prefetched.  Byte positions inside a step come from a wave scan of the lengths; each value's <= 10 bytes are
// shifted to their byte phase and OR-ed into the wave's zeroed LDS tile (neighbours share boundary dwords), and the
// tile leaves as aligned 16-byte stores.  The up to 15 bytes that do not fill a 16-byte unit are carried into
// the next step's tile, so every global store but the row's last few bytes is a full aligned unit.  The row's byte
// count is only known at the end, hence the slotted output (row r at r * slot): each vector is sealed on its own
// anyway (sodium.rs:36-43), so contiguity of the rows has no meaning on the wire.
static constexpr int kEncVals = 128;
static constexpr int kEncTile = 16 + kEncVals * 10 + 32;      // carry | bytes | slack for the last value's dwords

__device__ __forceinline__ void value_bytes(uint64_t zz, uint32_t len, uint32_t& b0, uint32_t& b1, uint32_t& b2) {
    uint32_t d0 = (uint32_t)zz & 0x0FFFFFFFu, d1 = (uint32_t)(zz >> 28) & 0x0FFFFFFFu;   // 2 x 28 bits -> 2 x 4 bytes
    d0 = (d0 & 0x00003FFFu) | ((d0 & 0x0FFFC000u) << 2);
    d1 = (d1 & 0x00003FFFu) | ((d1 & 0x0FFFC000u) << 2);
    d0 = (d0 & 0x007F007Fu) | ((d0 & 0x3F803F80u) << 1);
    d1 = (d1 & 0x007F007Fu) | ((d1 & 0x3F803F80u) << 1);
    const uint32_t d2 = ((uint32_t)(zz >> 56) & 0x7Fu) | ((uint32_t)(zz >> 63) << 8);     // bytes 8 and 9
    // continuation bit on every byte but the last one
    const uint64_t cont = len >= 9 ? 0x8080808080808080ull : (0x0080808080808080ull >> (8 * ((8 - len) & 7)));   // len 0: unused
    b0 = d0 | (uint32_t)cont;
    b1 = d1 | (uint32_t)(cont >> 32);
    b2 = d2 | (len > 9 ? 0x80u : 0u);
    if (len == 0) { b0 = 0; b1 = 0; b2 = 0; }
}

__device__ __forceinline__ void tile_or(uint32_t* tile32, uint32_t pos, uint32_t b0, uint32_t b1, uint32_t b2) {
    const uint32_t sh = 8u * (pos & 3u);
    const uint64_t lo = ((uint64_t)b1 << 32 | b0) << sh;                 // bytes 0..7 at their phase (low part)
    const uint64_t hi = (((uint64_t)b2 << 32) | b1) << sh;               // high dword: b2 and what left b1
    uint32_t* t = tile32 + (pos >> 2);
    atomicOr(t + 0, (uint32_t)lo);
    atomicOr(t + 1, (uint32_t)(lo >> 32));
    atomicOr(t + 2, (uint32_t)(hi >> 32));
    if (sh) atomicOr(t + 3, (uint32_t)(((uint64_t)b2 << sh) >> 32));
}

__device__ __forceinline__ void enc_load2(const int64_t* __restrict__ src, uint64_t len, bool vec, uint64_t i, int64_t& x,
                                          int64_t& y) {
    x = 0; y = 0;
    if (i + 1 < len && vec) {
        typedef long long ll2v __attribute__((ext_vector_type(2)));
        const ll2v v = __builtin_nontemporal_load(reinterpret_cast<const ll2v*>(src + i));
        x = v.x; y = v.y;
    } else {
        if (i < len) x = src[i];
        if (i + 1 < len) y = src[i + 1];
    }
}

__global__ __launch_bounds__(kStreamWaves * 64) void varint_stream_encode_kernel(VarintRows R, uint8_t* __restrict__ out,
                                                                                 uint64_t slot,
                                                                                 uint64_t* __restrict__ row_bytes) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[kStreamWaves][kEncTile];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * kStreamWaves + wave;
    if (r >= R.rows) return;
    const int64_t* __restrict__ src = R.values + r * R.row_stride;
    uint8_t* __restrict__ dst = out + r * slot;
    uint32_t* tile32 = reinterpret_cast<uint32_t*>(tiles[wave]);
    uint4* tile128 = reinterpret_cast<uint4*>(tiles[wave]);
    const bool vec = (((uintptr_t)src) & 15u) == 0;
    const uint64_t n_steps = (R.len + kEncVals - 1) / kEncVals;
    const uint64_t len = R.len;
    int64_t wx[kStreamDepth], wy[kStreamDepth];
#pragma unroll
    for (int d = 0; d < kStreamDepth; ++d) enc_load2(src, len, vec, (uint64_t)d * kEncVals + 2 * (uint64_t)lane, wx[d], wy[d]);
    uint64_t cur = 0;                 // bytes of this row already in global memory (multiple of 16)
    uint32_t cb = 0;                  // carried bytes (< 16), held by lane 0 in `carry`
    uint4 carry = make_uint4(0, 0, 0, 0);
    for (uint64_t j0 = 0; j0 < n_steps; j0 += kStreamDepth) {
#pragma unroll
        for (int d = 0; d < kStreamDepth; ++d) {
            const uint64_t j = j0 + d;
            if (j >= n_steps) continue;
            int64_t x = wx[d], y = wy[d];
            asm volatile("" : "+v"(x), "+v"(y));
            __builtin_amdgcn_sched_barrier(0);
            enc_load2(src, len, vec, (j + kStreamDepth) * kEncVals + 2 * (uint64_t)lane, wx[d], wy[d]);
            __builtin_amdgcn_sched_barrier(0);
            const uint64_t i = j * kEncVals + 2 * (uint64_t)lane;
            const uint64_t zx = zigzag(x), zy = zigzag(y);
            const uint32_t lx = i < len ? varint_len(zx) : 0u, ly = i + 1 < len ? varint_len(zy) : 0u;
            const uint32_t cnt = lx + ly;
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t T = __builtin_amdgcn_readlane(incl, 63);
            const uint32_t pos = cb + incl - cnt;
            // zero the tile, put the carried bytes in front
            const bool first = lane == 0;
            tile128[lane] = make_uint4(first ? carry.x : 0u, first ? carry.y : 0u, first ? carry.z : 0u, first ? carry.w : 0u);
            if (lane + 64 < kEncTile / 16) tile128[lane + 64] = make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t b0, b1, b2;
            value_bytes(zx, lx, b0, b1, b2);
            tile_or(tile32, pos, b0, b1, b2);
            value_bytes(zy, ly, b0, b1, b2);
            tile_or(tile32, pos + lx, b0, b1, b2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t have = cb + T, units = have >> 4;              // <= 81 units
            uint4* g = reinterpret_cast<uint4*>(dst + cur);
            if ((uint32_t)lane < units) g[lane] = tile128[lane];
            if ((uint32_t)lane + 64 < units) g[lane + 64] = tile128[lane + 64];
            const uint4 next = tile128[units];                            // bytes past `have` are zero
            carry = make_uint4(__builtin_amdgcn_readfirstlane(next.x), __builtin_amdgcn_readfirstlane(next.y),
                               __builtin_amdgcn_readfirstlane(next.z), __builtin_amdgcn_readfirstlane(next.w));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cur += (uint64_t)units * 16;
            cb = have & 15u;
        }
    }
    // the last bytes of the row
    if ((uint32_t)lane < cb) {
        const uint32_t wsel = lane >> 2;
        const uint32_t word = wsel == 0 ? carry.x : wsel == 1 ? carry.y : wsel == 2 ? carry.z : carry.w;
        dst[cur + lane] = (uint8_t)(word >> (8 * (lane & 3)));
    }
    if (lane == 0) row_bytes[r] = cur + cb;
}

// ---- wire format -> clerk sums without the decoded tile ---------------------------------
// The 16 (or 8) waves of a workgroup stream as many rows of ONE job in lockstep (a group of kStreamDepth chunks each, then a
// barrier).  Values go into a sliding window of kCombWindow columns in LDS, two 64-bit planes per column (sum of
// the low 32 bits, sum of the arithmetic high 32 bits: no carries, plain ds_add_u64).  After every group the columns
// below min(current column of the 16 rows) can no longer be touched: they are folded into (lo, hi) and added to
// the global 128-bit accumulators, one pair of atomics per column and workgroup.  A value beyond the window (rows
// drifting apart by more than ~1500 columns: only with wildly different value sizes) goes to the global
// accumulator directly, so the result is exact for any input.
static constexpr int kCombWindow = 2048;

struct WindowSink {
    unsigned long long* lo32; unsigned long long* hi32;   // LDS planes
    uint64_t base;                                         // first column of the window
    uint64_t* acc_lo; int64_t* acc_hi;                     // this job's accumulators
    __device__ __forceinline__ void operator()(uint64_t c, int64_t v) const {
        if (c - base < (uint64_t)kCombWindow) {
            const uint32_t i = (uint32_t)c & (kCombWindow - 1);
            atomicAdd(&lo32[i], (unsigned long long)((uint64_t)v & 0xFFFFFFFFull));
            atomicAdd(&hi32[i], (unsigned long long)(v >> 32));
        } else {
            acc_atomic_add(acc_lo + c, acc_hi + c, (uint64_t)v, v >> 63);
        }
    }
};

template <int kCombWaves>       // rows per workgroup: 16 (one pair of global atomics per column and 16 rows), 8 when rows are few
__global__ __launch_bounds__(kCombWaves * 64) void varint_stream_combine_kernel(
    const uint8_t* __restrict__ bytes, uint64_t n_bytes, RowRanges offsets, uint64_t rows_per_job,
    uint64_t groups_per_job, uint64_t len, uint64_t* __restrict__ acc_lo, int64_t* __restrict__ acc_hi,
    uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[kCombWaves][kStreamTile];
    __shared__ unsigned long long lo32[kCombWindow], hi32[kCombWindow];
    __shared__ uint64_t cols[kCombWaves];
    const int wave = threadIdx.x >> 6;
    const uint64_t job = blockIdx.x / groups_per_job, grp = blockIdx.x - job * groups_per_job;
    const uint64_t rj = grp * kCombWaves + wave;                  // row inside the job
    for (int i = threadIdx.x; i < kCombWindow; i += kCombWaves * 64) { lo32[i] = 0; hi32[i] = 0; }
    RowStream rs;
    bool live = false;
    if (rj < rows_per_job) {
        uint64_t a, b;
        live = stream_row_range(bytes, n_bytes, offsets, job * rows_per_job + rj, len, status, a, b);
        if (live) rs.open(bytes, a, b, len, tiles[wave]);
    }
    WindowSink sink{lo32, hi32, 0, acc_lo + job * len, acc_hi + job * len};
    __syncthreads();
    for (;;) {
        if (live) {
            rs.next_group(sink);
            if (rs.done()) { rs.close(status); live = false; }
        }
        if ((threadIdx.x & 63) == 0) cols[wave] = live ? rs.col : len;     // a finished row no longer holds the window
        __syncthreads();
        uint64_t nb = len;
#pragma unroll
        for (int w = 0; w < kCombWaves; ++w) nb = cols[w] < nb ? cols[w] : nb;
        // every later value has column >= nb: fold and flush [base, nb)
        uint64_t end = nb < sink.base + kCombWindow ? nb : sink.base + kCombWindow;
        for (uint64_t c = sink.base + threadIdx.x; c < end; c += kCombWaves * 64) {
            const uint32_t i = (uint32_t)c & (kCombWindow - 1);
            const uint64_t A = lo32[i];
            const int64_t Bq = (int64_t)hi32[i];
            if (A | (uint64_t)Bq) {
                lo32[i] = 0; hi32[i] = 0;
                const uint64_t lo = A + ((uint64_t)Bq << 32);
                acc_atomic_add(sink.acc_lo + c, sink.acc_hi + c, lo, (Bq >> 32) + (lo < A ? 1 : 0));
            }
        }
        sink.base = nb;
        __syncthreads();
        if (nb >= len) break;                                     // all 16 rows are through
    }
}

// ---- launchers ------------------------------------------------------------------------------------
static inline uint64_t vceil(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

size_t varint_encode_blocks(size_t rows, size_t len) { return (size_t)vceil((uint64_t)rows * len, kVT * kVals); }
size_t varint_decode_blocks(size_t n_bytes) { return (size_t)vceil(n_bytes, kVT * kBytes); }

hipError_t launch_varint_lengths(const VarintRows& R, uint32_t* d_block_bytes, hipStream_t s) {
    const size_t nb = varint_encode_blocks(R.rows, R.len);
    if (nb == 0) return hipSuccess;
    if (nb > 0xFFFFFFFFull / kVT) return hipErrorInvalidConfiguration;   // < 2^32 work-items per launch
    varint_len_kernel<<<dim3((unsigned)nb), dim3(kVT), 0, s>>>(R, d_block_bytes);
    return hipGetLastError();
}

size_t scan_aux_entries(size_t n) { return (n + 1023) / 1024 + 1; }

hipError_t launch_scan_u32(const uint32_t* d_in, uint64_t* d_out, size_t n, uint64_t* d_total, uint64_t* d_aux,
                           hipStream_t s) {
    const size_t m = (n + 1023) / 1024;
    if (m == 0) return hipMemsetAsync(d_total, 0, 8, s);
    if (m > 0xFFFFFFFFull / 1024) return hipErrorInvalidConfiguration;
    scan_chunks_kernel<<<dim3((unsigned)m), dim3(1024), 0, s>>>(d_in, d_out, d_aux, n);
    scan_totals_kernel<<<dim3(1), dim3(1024), 0, s>>>(d_aux, m, d_total);
    scan_add_kernel<<<dim3((unsigned)m), dim3(1024), 0, s>>>(d_out, d_aux, n);
    return hipGetLastError();
}

hipError_t launch_varint_write(const VarintRows& R, const uint64_t* d_block_off, uint8_t* d_out, uint64_t* d_row_offsets,
                               hipStream_t s) {
    const size_t nb = varint_encode_blocks(R.rows, R.len);
    if (nb == 0) return hipSuccess;
    varint_write_kernel<<<dim3((unsigned)nb), dim3(kVT), 0, s>>>(R, d_block_off, d_out, d_row_offsets);
    return hipGetLastError();
}

hipError_t launch_varint_count(const uint8_t* d_bytes, size_t n_bytes, uint32_t* d_block_counts, hipStream_t s) {
    const size_t nb = varint_decode_blocks(n_bytes);
    if (nb == 0) return hipSuccess;
    if (nb > 0xFFFFFFFFull / kVT) return hipErrorInvalidConfiguration;
    varint_count_kernel<<<dim3((unsigned)nb), dim3(kVT), 0, s>>>(d_bytes, n_bytes, d_block_counts);
    return hipGetLastError();
}

hipError_t launch_varint_decode(const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_block_val_off, size_t rows,
                                size_t len, size_t row_stride, int64_t* d_out, uint32_t* d_status, hipStream_t s) {
    const size_t nb = varint_decode_blocks(n_bytes);
    if (nb == 0) return hipSuccess;
    varint_decode_kernel<<<dim3((unsigned)nb), dim3(kVT), 0, s>>>(d_bytes, n_bytes, d_block_val_off, rows, len, row_stride,
                                                                  d_out, d_status);
    return hipGetLastError();
}

hipError_t launch_varint_stream_decode(const uint8_t* d_bytes, size_t n_bytes, const RowRanges& d_offsets, size_t rows,
                                       size_t len, size_t row_stride, int64_t* d_out, uint32_t* d_status, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    varint_stream_decode_kernel<<<dim3((unsigned)vceil(rows, kStreamWaves)), dim3(kStreamWaves * 64), 0, s>>>(
        d_bytes, n_bytes, d_offsets, rows, len, row_stride, d_out, d_status);
    return hipGetLastError();
}

hipError_t launch_varint_stream_encode(const VarintRows& R, uint8_t* d_out, size_t slot_bytes, uint64_t* d_row_bytes,
                                       hipStream_t s) {
    if (R.rows == 0) return hipSuccess;
    varint_stream_encode_kernel<<<dim3((unsigned)vceil(R.rows, kStreamWaves)), dim3(kStreamWaves * 64), residency_pad_bytes(knob(KNOB_WIRE_WG_PER_CU), 8192), s>>>(
        R, d_out, slot_bytes, d_row_bytes);
    return hipGetLastError();
}

hipError_t launch_varint_stream_combine(const uint8_t* d_bytes, size_t n_bytes, const RowRanges& d_offsets, size_t jobs,
                                        size_t rows_per_job, size_t len, uint64_t* d_acc_lo, int64_t* d_acc_hi,
                                        uint32_t* d_status, hipStream_t s) {
    if (jobs == 0 || rows_per_job == 0 || len == 0) return hipSuccess;
    // 16 rows per workgroup once that still gives every CU two workgroups, else 8 (more workgroups, twice the atomics)
    const bool wide = vceil(rows_per_job, 16) * jobs >= 512;
    const uint64_t groups = vceil(rows_per_job, wide ? 16 : 8);
    if (groups * jobs > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    if (wide)
        varint_stream_combine_kernel<16><<<dim3((unsigned)(groups * jobs)), dim3(16 * 64), residency_pad_bytes(knob(KNOB_WIRE_WG_PER_CU), 49152), s>>>(
            d_bytes, n_bytes, d_offsets, rows_per_job, groups, len, d_acc_lo, d_acc_hi, d_status);
    else
        varint_stream_combine_kernel<8><<<dim3((unsigned)(groups * jobs)), dim3(8 * 64), 0, s>>>(
            d_bytes, n_bytes, d_offsets, rows_per_job, groups, len, d_acc_lo, d_acc_hi, d_status);
    return hipGetLastError();
}

hipError_t launch_varint_rowcheck(const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_offsets, size_t rows, size_t len,
                                  const uint64_t* d_block_val_off, uint32_t* d_status, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    varint_rowcheck_kernel<<<dim3((unsigned)vceil(rows, kVT / 64)), dim3(kVT), 0, s>>>(d_bytes, n_bytes, d_offsets, rows, len,
                                                                                   d_block_val_off, d_status);
    return hipGetLastError();
}

}  // namespace sda
