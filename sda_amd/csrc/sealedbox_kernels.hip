// Batch open / seal of libsodium sealed boxes on gfx950: what the reference does one payload at a
// time through sodiumoxide - `sealedbox::seal` per clerk in participate.rs:82-101 via encryption/sodium.rs:43,
// `sealedbox::open` for each of the P encryptions of a clerking job in clerk.rs:79-82 via sodium.rs:78.
//
//   sbox_setup_kernel   one DPP quad = one X25519 ladder (open: one per box, seal: two side by side); its first lane then
//                       derives the BLAKE2b nonce, two HSalsa20, the Poly1305 key and its powers
//   sbox_stream_kernel  one lane = one 64-byte Salsa20 block: XSalsa20 keystream xor (VALU-bound: 20 rounds per 64 B)
//   sbox_poly_kernel    one wave = one 16 KiB region of ciphertext: lane-strided Horner with the uniform multiplier
//                       r^64 (one 130-bit multiply per 16-byte piece, 1 KiB coalesced loads), lane weights r^(l+1)
//   sbox_final_kernel   one lane = one box: Horner over the regions with r^1024, tag compare (open) / tag store (seal)
//
// Pieces are indexed by their distance d from the END of the message (tag = sum c_d r^d + s), so the only ragged
// region is the one at the start of the message, where missing pieces simply contribute nothing.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "capi_internal.hpp"
#include "kernels.hpp"
#include "sbox_primitives.hpp"

namespace sda {

using namespace sbx;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr int kSbThreads = 256;
static constexpr int kPolySteps = 16;                       // pieces per lane and region: region = 64 * 16 pieces = 16 KiB

struct SboxKeyArg { uint32_t w[8]; };

__device__ __forceinline__ void load_words(uint32_t* w, const uint8_t* p, int n) {       // p is 4-byte aligned
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    for (int i = 0; i < n; ++i) w[i] = q[i];
}

// common tail of setup: from the shared X25519 secret and the nonce to the per-box state
__device__ __forceinline__ void derive_state(SboxState& st, const uint32_t shared[8], const uint32_t nonce[6]) {
    const uint32_t zero4[4] = {0, 0, 0, 0};
    uint32_t k[8];
    hsalsa20(k, shared, zero4);                               // crypto_box_beforenm
    hsalsa20(st.subkey, k, nonce);                            // XSalsa20: subkey from the first 16 nonce bytes
    st.n0 = nonce[4]; st.n1 = nonce[5];
    uint32_t b0[16];
    salsa20_block(b0, st.subkey, st.n0, st.n1, 0);            // stream bytes 0..31 = the one-time Poly1305 key
    P26 r, rp;
    p26_clamped_r(r, b0);
#pragma unroll
    for (int i = 0; i < 4; ++i) st.s[i] = b0[4 + i];
    rp = r;
    for (int i = 0; i < 64; ++i) {                            // rpow[i] = r^(i+1)
#pragma unroll
        for (int j = 0; j < 5; ++j) st.rpow[i][j] = rp.v[j];
        if (i < 63) { P26 t; p26_mul(t, rp, r); rp = t; }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) st.r64[j] = rp.v[j];
    for (int s = kPolySteps; s > 1; s >>= 1) { P26 t; p26_mul(t, rp, rp); rp = t; }     // (r^64)^16 = r^1024
#pragma unroll
    for (int j = 0; j < 5; ++j) st.rS[j] = rp.v[j];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) any |= shared[i];
    st.bad = any == 0 ? 1u : 0u;                              // all-zero shared secret (small-order point): crypto_box refuses it
}

// ---- X25519 across the four lanes of a DPP quad ---------------------------------------------------------------------
// A box's setup is one lane's worth of strictly sequential work (255 ladder steps of 5 multiplications + 4 squarings in
// GF(2^255 - 19)) and a job has only thousands of boxes, so the ladder's LATENCY is what the caller waits for.  A ladder
// step (RFC 7748 section 5) is three levels of mutually independent products:
//     level 1:  AA = A * A      BB = B * B      DA = D * A      CB = C * B
//     level 2:  x3 = (DA+CB)^2  t = (DA-CB)^2   x2 = AA * BB    u = a24 * E          (E = AA - BB)
//     level 3:  z3 = x1 * t                     z2 = E * (AA + u)
// Lane c of the quad computes the c-th product of each level - one uniform fe_mul(P, Q) with lane-selected operands -
// and the products are handed round with v_mov_b32_dpp quad_perm broadcasts; every lane keeps the whole ladder state
// (x2, z2, x3, z3), so additions, the conditional swap and the final inversion need no exchange.  Three multiplications
// per step on the critical path instead of ten.  All four lanes of the quad must be active and pass the same k and u;
// all four return the same result.
template <int SRC>
__device__ __forceinline__ void fe_quad_bcast(Fe& h, const Fe& f) {
#pragma unroll
    for (int i = 0; i < 10; ++i) h.v[i] = __builtin_amdgcn_mov_dpp(f.v[i], SRC * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ void fe_select(Fe& h, bool c, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < 10; ++i) h.v[i] = c ? a.v[i] : b.v[i];
}

__device__ __noinline__ void x25519_quad(uint32_t out[8], const uint32_t k_in[8], const uint32_t u[8]) {
    const uint32_t c = threadIdx.x & 3u;
    const bool c0 = c == 0, c1 = c == 1, c2 = c == 2, lo = c < 2;
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = k_in[i];
    k[0] &= 0xFFFFFFF8u;
    k[7] = (k[7] & 0x7FFFFFFFu) | 0x40000000u;
    Fe x1, x2, z2, x3, z3, a24;
    fe_from_words(x1, u);
    fe_set(x2, 1); fe_set(z2, 0); x3 = x1; fe_set(z3, 1);
    fe_set(a24, 121665);
    uint32_t swap = 0;
    for (int t = 254; t >= 0; --t) {
        const uint32_t kt = (k[t >> 5] >> (t & 31)) & 1u;
        swap ^= kt;
        fe_cswap(x2, x3, swap);
        fe_cswap(z2, z3, swap);
        swap = kt;
        Fe A, B, C, D, P, Q, T, r;
        fe_add(A, x2, z2); fe_sub(B, x2, z2); fe_add(C, x3, z3); fe_sub(D, x3, z3);
        fe_select(T, c2, D, C); fe_select(P, c1, B, T); fe_select(P, c0, A, P);      // A | B | D | C
        fe_select(Q, (c & 1u) != 0, B, A);                                           // A | B | A | B
        fe_mul(r, P, Q);
        Fe AA, BB, DA, CB, E, S, Df;
        fe_quad_bcast<0>(AA, r); fe_quad_bcast<1>(BB, r); fe_quad_bcast<2>(DA, r); fe_quad_bcast<3>(CB, r);
        fe_sub(E, AA, BB); fe_add(S, DA, CB); fe_sub(Df, DA, CB);
        fe_select(T, c2, AA, E); fe_select(P, c1, Df, T); fe_select(P, c0, S, P);    // S | Df | AA | E
        fe_select(T, c2, BB, a24); fe_select(Q, lo, P, T);                           // S | Df | BB | a24
        fe_mul(r, P, Q);
        Fe tt, uu, W;
        fe_quad_bcast<0>(x3, r); fe_quad_bcast<1>(tt, r); fe_quad_bcast<2>(x2, r); fe_quad_bcast<3>(uu, r);
        fe_add(W, AA, uu);
        fe_select(P, lo, x1, E);
        fe_select(Q, lo, tt, W);
        fe_mul(r, P, Q);
        fe_quad_bcast<0>(z3, r); fe_quad_bcast<2>(z2, r);
    }
    fe_cswap(x2, x3, swap);
    fe_cswap(z2, z3, swap);
    Fe zi;
    fe_invert(zi, z2);
    fe_mul(x2, x2, zi);
    fe_to_words(out, x2);
}

// OPEN: box r at boxes + r * slot (epk || tag || ciphertext), recipient key pair in the kernel arguments; four lanes per box
__global__ __launch_bounds__(64) void sbox_setup_open_kernel(const uint8_t* __restrict__ boxes, size_t slot,
                                                             const uint64_t* __restrict__ row_bytes, size_t rows, SboxKeyArg pk,
                                                             SboxKeyArg sk, SboxState* __restrict__ states) {
    const size_t r = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 2;
    if (r >= rows) return;                                    // whole quads leave together
    const bool lead = (threadIdx.x & 3u) == 0;
    SboxState& st = states[r];
    if (row_bytes[r] < 48) { if (lead) st.bad = 1; return; }
    uint32_t epk[8], shared[8];
    load_words(epk, boxes + r * slot, 8);
    x25519_quad(shared, sk.w, epk);
    if (lead) {
        uint32_t nonce[6];
        seal_nonce(nonce, epk, pk.w);
        derive_state(st, shared, nonce);
    }
}

// SEAL: ephemeral secret r at esk + 32 r; recipient key of row r = pks[(r / rows_per_key) % n_pks]; writes epk to the box.
// Eight lanes per box: the two ladders of a seal (esk * base point -> epk, esk * pk -> shared secret) share the scalar and
// run side by side in the two quads.
__global__ __launch_bounds__(64) void sbox_setup_seal_kernel(const uint8_t* __restrict__ esk, const uint8_t* __restrict__ pks,
                                                             size_t n_pks, size_t rows_per_key, uint8_t* __restrict__ boxes,
                                                             size_t slot, size_t rows, SboxState* __restrict__ states) {
    const size_t r = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 3;
    if (r >= rows) return;                                    // whole groups of eight leave together
    const bool second = (threadIdx.x & 4u) != 0;
    uint32_t e[8], pk[8], point[8], res[8], other[8];
    load_words(e, esk + 32 * r, 8);
    load_words(pk, pks + 32 * ((r / rows_per_key) % n_pks), 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) point[i] = second ? pk[i] : (i == 0 ? 9u : 0u);
    x25519_quad(res, e, point);
#pragma unroll
    for (int i = 0; i < 8; ++i) other[i] = (uint32_t)__shfl_down((int)res[i], 4);   // the second quad's result: the shared secret
    if ((threadIdx.x & 7u) == 0) {
        uint32_t* o = reinterpret_cast<uint32_t*>(boxes + r * slot);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = res[i];                                  // epk
        uint32_t nonce[6];
        seal_nonce(nonce, res, pk);
        derive_state(states[r], other, nonce);
    }
}

// XSalsa20 xor: in row r at in + r * in_slot + in_off, out likewise; the message of row r has lens[r] - len_sub bytes.
// One lane computes one 64-byte Salsa20 block (block J covers message bytes [64 J - 32, 64 J + 32): four whole 16-byte
// pieces); the wave's 64 blocks go through a transposed LDS tile so that the global loads and stores are lane-contiguous
// 16-byte pieces (1 KiB per wave-instruction) instead of 64-byte-strided ones.
__global__ __launch_bounds__(kSbThreads) void sbox_stream_kernel(const uint8_t* __restrict__ in, size_t in_slot, size_t in_off,
                                                                 uint8_t* __restrict__ out, size_t out_slot, size_t out_off,
                                                                 const uint64_t* __restrict__ lens, uint64_t len_sub,
                                                                 uint64_t max_msg, const SboxState* __restrict__ states, size_t row0) {
    __shared__ uint32_t tile[kSbThreads / 64][16][65];                      // [wave][word][block], padded rows
    const size_t r = row0 + blockIdx.y;
    const uint64_t have = lens[r];
    // a row longer than the caller's bound is refused, never read; a row marked bad - small-order recipient key / all-zero
    // shared secret (seal and open), or a box that failed authentication (open: the verdict is in before this pass runs) -
    // is neither encrypted under the degenerate key nor decrypted into the caller's buffer
    const bool live = have >= len_sub && have - len_sub <= max_msg && !states[r].bad;
    const uint64_t mlen = live ? have - len_sub : 0;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t J0 = (uint64_t)blockIdx.x * kSbThreads + (uint64_t)wave * 64;   // first Salsa20 block of this wave
    const uint64_t J = J0 + lane;
    const uint64_t a = J ? 64 * J - 32 : 0;
    if (a < mlen) {
        const SboxState& st = states[r];
        uint32_t key[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) key[i] = st.subkey[i];                  // wave-uniform: scalar loads
        uint32_t ks[16];
        salsa20_block(ks, key, st.n0, st.n1, J);
#pragma unroll
        for (int w = 0; w < 16; ++w) tile[wave][w][lane] = ks[w];
    }
    __syncthreads();
    if (mlen == 0) return;
    const uint8_t* src = in + r * in_slot + in_off;
    uint8_t* dst = out + r * out_slot + out_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t sp = lane + 64 * i;                                  // 16-byte piece of the wave's 4 KiB of key stream
        const uint64_t spos = 64 * J0 + 16 * (uint64_t)sp;                  // its stream offset; message offset = stream offset - 32
        if (spos < 32) continue;                                            // stream bytes 0..31 are the Poly1305 key
        const uint64_t m = spos - 32;
        if (m >= mlen) continue;
        const uint32_t blk = sp >> 2, w0 = (sp & 3) * 4;
        const uint32_t k0 = tile[wave][w0][blk], k1 = tile[wave][w0 + 1][blk], k2 = tile[wave][w0 + 2][blk], k3 = tile[wave][w0 + 3][blk];
        const uint64_t n = mlen - m;
        if (n >= 16) {
            uint4 v = *reinterpret_cast<const uint4*>(src + m);
            v.x ^= k0; v.y ^= k1; v.z ^= k2; v.w ^= k3;
            *reinterpret_cast<uint4*>(dst + m) = v;
        } else {
            const uint32_t kk[4] = {k0, k1, k2, k3};
            for (uint32_t bb = 0; bb < (uint32_t)n; ++bb) {
                uint32_t word = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) word = (bb >> 2) == (uint32_t)q ? kk[q] : word;
                dst[m + bb] = src[m + bb] ^ (uint8_t)(word >> (8 * (bb & 3)));
            }
        }
    }
}

// Poly1305 partial sums over the ciphertext (row r at ct + r * slot + off, lens[r] - len_sub bytes)
__global__ __launch_bounds__(kSbThreads) void sbox_poly_kernel(const uint8_t* __restrict__ ct, size_t slot, size_t off,
                                                               const uint64_t* __restrict__ lens, uint64_t len_sub, uint64_t max_msg,
                                                               const SboxState* __restrict__ states, uint32_t* __restrict__ partial,
                                                               size_t regions, size_t row0) {
    const size_t r = row0 + blockIdx.y;
    const uint32_t lane = threadIdx.x & 63;
    const size_t region = (size_t)blockIdx.x * (kSbThreads / 64) + (threadIdx.x >> 6);
    if (region >= regions) return;
    const uint64_t have = lens[r];
    const uint64_t mlen = (have >= len_sub && have - len_sub <= max_msg) ? have - len_sub : 0;   // over-long rows: refused in final
    const uint64_t npieces = (mlen + 15) / 16;
    const uint32_t tail = (uint32_t)(mlen & 15);                             // bytes of the last piece (0 = full)
    const SboxState& st = states[r];
    P26 R64;
#pragma unroll
    for (int j = 0; j < 5; ++j) R64.v[j] = st.r64[j];                        // wave-uniform
    const uint8_t* base = ct + r * slot + off;
    P26 h;
#pragma unroll
    for (int j = 0; j < 5; ++j) h.v[j] = 0;
    const uint64_t d0 = (uint64_t)region * 64 * kPolySteps + lane + 1;       // distance from the end, step 0
    if (d0 - lane <= npieces) {                                              // the region holds at least one piece (wave-uniform)
        for (int m = kPolySteps - 1; m >= 0; --m) {
            const uint64_t d = d0 + 64 * (uint64_t)m;
            P26 t;
            p26_mul(t, h, R64);
            h = t;
            if (d <= npieces) {
                const uint64_t b = npieces - d;
                uint32_t w[4];
                const uint32_t nb = (d == 1 && tail) ? tail : 16u;
                if (nb == 16) {
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + 16 * b));
                    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
                } else {
                    w[0] = w[1] = w[2] = w[3] = 0;
                    for (uint32_t i = 0; i < nb; ++i) w[i >> 2] |= (uint32_t)base[16 * b + i] << (8 * (i & 3));
                }
                P26 c;
                p26_from_piece(c, w, nb);
                p26_add(h, h, c);
            }
        }
        P26 wgt, t;
#pragma unroll
        for (int j = 0; j < 5; ++j) wgt.v[j] = st.rpow[lane][j];             // r^(lane+1)
        p26_mul(t, h, wgt);
        h = t;
        p26_carry(h);                                                        // limbs < 2^26: 64 of them sum below 2^32
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        uint32_t v = h.v[j];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
        h.v[j] = v;
    }
    if (lane == 0) {
        uint32_t* o = partial + (r * regions + region) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = h.v[j];
    }
}

// one lane per box: total = sum over regions of partial[region] * (r^1024)^region, tag = total + s.
// OPEN (seal == 0): compare with the tag in the box; out_bytes[r] = message length, or 0 for a box that fails.
// SEAL: store the tag, row_bytes_out[r] = message length + 48.
__global__ __launch_bounds__(64) void sbox_final_kernel(const uint32_t* __restrict__ partial, size_t regions,
                                                        SboxState* __restrict__ states, uint8_t* __restrict__ boxes, size_t slot,
                                                        const uint64_t* __restrict__ lens, uint64_t len_sub, uint64_t max_msg, size_t rows, int seal,
                                                        uint64_t* __restrict__ out_bytes, uint32_t* __restrict__ ok,
                                                        uint32_t* __restrict__ status) {
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    SboxState& st = states[r];
    const uint64_t have = lens[r];
    bool good = have >= len_sub && have - len_sub <= max_msg && !st.bad;    // longer than the launch was sized for: refused
    const uint64_t mlen = good ? have - len_sub : 0;
    uint32_t tag[4] = {0, 0, 0, 0};
    if (good) {
        const size_t used = (size_t)(((mlen + 15) / 16 + 64 * kPolySteps - 1) / (64 * kPolySteps));
        P26 RS, acc;
#pragma unroll
        for (int j = 0; j < 5; ++j) { RS.v[j] = st.rS[j]; acc.v[j] = 0; }
        for (size_t g = used; g-- > 0;) {
            P26 t, p;
            p26_mul(t, acc, RS);
#pragma unroll
            for (int j = 0; j < 5; ++j) p.v[j] = partial[(r * regions + g) * 5 + j];
            p26_carry(p);
            p26_add(acc, t, p);
        }
        p26_finish(tag, acc, st.s);
    }
    uint32_t* box_tag = reinterpret_cast<uint32_t*>(boxes + r * slot + 32);
    if (seal) {
        if (good) {
#pragma unroll
            for (int i = 0; i < 4; ++i) box_tag[i] = tag[i];
        }
        out_bytes[r] = good ? mlen + 48 : 0;                   // a message longer than the launch was sized for is not sealed
        return;
    }
    if (good) {                                                // the tag is only read from a box that has one
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) diff |= tag[i] ^ box_tag[i];
        good = diff == 0;
    }
    out_bytes[r] = good ? mlen : 0;
    if (ok) ok[r] = good ? 1u : 0u;
    if (!good) { atomicOr(status, 16u); states[r].bad = 1; }   // the keystream pass that follows leaves this row's output alone
}

// ---- launchers ---------------------------------------------------------------------------------------------
static inline uint64_t cdiv64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
size_t sbox_regions(size_t max_msg_bytes) { return (size_t)cdiv64(cdiv64(max_msg_bytes, 16), 64 * kPolySteps) + (max_msg_bytes == 0 ? 1 : 0); }

static SboxKeyArg key_arg(const uint8_t k[32]) {
    SboxKeyArg a;
    for (int i = 0; i < 8; ++i) a.w[i] = (uint32_t)k[4 * i] | ((uint32_t)k[4 * i + 1] << 8) | ((uint32_t)k[4 * i + 2] << 16) | ((uint32_t)k[4 * i + 3] << 24);
    return a;
}

enum SboxPass { kPassStream = 1, kPassPoly = 2 };

// the bulk passes over the rows' payloads: `first` then `second` (each kPassStream, kPassPoly or 0)
static hipError_t bulk(const uint8_t* d_in, size_t in_slot, size_t in_off, uint8_t* d_out, size_t out_slot, size_t out_off,
                       const uint8_t* d_ct, size_t ct_slot, size_t ct_off, const uint64_t* d_lens, uint64_t len_sub, size_t rows,
                       size_t max_msg, const SboxState* d_states, uint32_t* d_partial, int first, int second, hipStream_t s) {
    const size_t regions = sbox_regions(max_msg);
    const uint64_t sblocks = cdiv64(cdiv64(max_msg + 32, 64), kSbThreads);
    const uint64_t pblocks = cdiv64(regions, kSbThreads / 64);
    if (sblocks > 0x7FFFFFFFull || pblocks > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    for (size_t r0 = 0; r0 < rows; r0 += 65535) {
        const unsigned nr = (unsigned)(rows - r0 < 65535 ? rows - r0 : 65535);
        for (int pass : {first, second}) {
            if (pass == kPassStream && max_msg)
                sbox_stream_kernel<<<dim3((unsigned)sblocks, nr), dim3(kSbThreads), residency_pad_bytes(knob(KNOB_SBOX_WG_PER_CU), 20480), s>>>(d_in, in_slot, in_off, d_out, out_slot, out_off,
                                                                                            d_lens, len_sub, max_msg, d_states, r0);
            if (pass == kPassPoly)
                sbox_poly_kernel<<<dim3((unsigned)pblocks, nr), dim3(kSbThreads), 0, s>>>(d_ct, ct_slot, ct_off, d_lens, len_sub, max_msg, d_states,
                                                                                          d_partial, regions, r0);
        }
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_sealedbox_open(const uint8_t pk[32], const uint8_t sk[32], const uint8_t* d_boxes, size_t slot,
                                 const uint64_t* d_row_bytes, size_t rows, size_t max_box_bytes, uint8_t* d_out, size_t out_slot,
                                 uint64_t* d_out_bytes, uint32_t* d_ok, uint32_t* d_status, SboxState* d_states,
                                 uint32_t* d_partial, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const size_t max_msg = max_box_bytes > 48 ? max_box_bytes - 48 : 0;
    SboxKeyArg apk = key_arg(pk), ask = key_arg(sk);
    sbox_setup_open_kernel<<<dim3((unsigned)cdiv64(4 * rows, 64)), dim3(64), 0, s>>>(d_boxes, slot, d_row_bytes, rows, apk, ask, d_states);
    volatile uint32_t* wipe = ask.w;
    for (int i = 0; i < 8; ++i) wipe[i] = 0;
    if (hipError_t e = hipGetLastError()) return e;
    // verify, THEN decrypt: the tag of every box is checked before the keystream pass runs, and that pass skips the rows that
    // failed - no unauthenticated plaintext ever reaches d_out (crypto_box_seal_open writes nothing on failure either)
    if (hipError_t e = bulk(d_boxes, slot, 48, d_out, out_slot, 0, d_boxes, slot, 48, d_row_bytes, 48, rows, max_msg, d_states,
                            d_partial, kPassPoly, 0, s))
        return e;
    sbox_final_kernel<<<dim3((unsigned)cdiv64(rows, 64)), dim3(64), 0, s>>>(d_partial, sbox_regions(max_msg), d_states,
                                                                           const_cast<uint8_t*>(d_boxes), slot, d_row_bytes, 48, max_msg, rows, 0,
                                                                           d_out_bytes, d_ok, d_status);
    if (hipError_t e = hipGetLastError()) return e;
    return bulk(d_boxes, slot, 48, d_out, out_slot, 0, d_boxes, slot, 48, d_row_bytes, 48, rows, max_msg, d_states, d_partial,
                kPassStream, 0, s);
}

hipError_t launch_sealedbox_seal(const uint8_t* d_esk, const uint8_t* d_pks, size_t n_pks, size_t rows_per_key,
                                 const uint8_t* d_msgs, size_t msg_slot, const uint64_t* d_msg_bytes, size_t rows,
                                 size_t max_msg_bytes, uint8_t* d_boxes, size_t slot, uint64_t* d_row_bytes, SboxState* d_states,
                                 uint32_t* d_partial, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    sbox_setup_seal_kernel<<<dim3((unsigned)cdiv64(8 * rows, 64)), dim3(64), 0, s>>>(d_esk, d_pks, n_pks, rows_per_key, d_boxes, slot, rows,
                                                                                d_states);
    if (hipError_t e = hipGetLastError()) return e;
    // encrypt into the box, then authenticate the ciphertext
    if (hipError_t e = bulk(d_msgs, msg_slot, 0, d_boxes, slot, 48, d_boxes, slot, 48, d_msg_bytes, 0, rows, max_msg_bytes, d_states,
                            d_partial, kPassStream, kPassPoly, s))
        return e;
    sbox_final_kernel<<<dim3((unsigned)cdiv64(rows, 64)), dim3(64), 0, s>>>(d_partial, sbox_regions(max_msg_bytes), d_states, d_boxes, slot,
                                                                           d_msg_bytes, 0, max_msg_bytes, rows, 1, d_row_bytes, nullptr, nullptr);
    return hipGetLastError();
}

}  // namespace sda
