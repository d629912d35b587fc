// Launch interface between the C ABI (sda_capi.cpp) and the gfx950 kernels (sda_kernels.hip).
// Plain structs and device pointers only.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>

namespace sda {

struct ModParams {
    uint64_t m;           // modulus, 2 <= m < 2^62
    uint64_t mu;          // floor(2^64 / m)          (Barrett)
    uint64_t lemire_thr;  // 2^64 mod m               (Lemire rejection threshold of the one-word-per-draw rule), for EVERY modulus
    uint64_t lemire_thr2; // 2^64 mod m^2             (threshold of the PAIRED rule, modarith.hpp) when drbg_paired(m), else 0: a
                          //                          field of its own, so that no helper can be handed the other rule's threshold
};

struct MontParams {
    uint64_t p;     // odd prime modulus
    uint64_t pinv;  // -p^{-1} mod 2^64
};

struct DrbgKey {
    uint32_t w[8];
};

// Montgomery-form matrix passed BY VALUE in the kernarg segment, so that rows are fetched with
// scalar loads (s_load) and the multiplier operands live in SGPRs.
#define SDA_MAT_ARG_MAX 448   // u64 entries (3.5 KiB of the 4 KiB kernarg budget)
struct MatArg {
    uint64_t e[SDA_MAT_ARG_MAX];
};

// Constants of the balanced-31-bit-limb Montgomery path (R = 2^62, radix B = 2^31); see
// packed_gen_l31_kernel.  MatArg entries are then (uint32)m0 | (uint64)(uint32)m1 << 32 with
// m1 * B + m0 = centred representative of (M_ji * 2^62 mod p), m0 in [-2^30, 2^30).
struct L31Params {
    uint64_t p;       // modulus
    uint64_t p2;      // 2p
    uint64_t h;       // (p + 1) / 2 : v >= h is centred to v - p
    int32_t  p0;      // p mod 2^31
    int32_t  p1;      // p >> 31
    uint32_t pinvB;   // -p^{-1} mod 2^31
    uint32_t wide;    // 1: the three-digit shapes of 9 .. 12 terms run their dot product as ONE group (admitted by the host on the
                      // actual constants of both share maps, l31_wide_group_ok); 2: ... in Karatsuba form, three multiply-adds per
                      // term (round 6; l31_karatsuba_ok admits the halves' cross columns); 0 everywhere else
    uint64_t np, np2; // 2^64 - p and 2^64 - 2p: x + np wraps exactly when x >= p (the conditional subtractions)
};

// Constants of the narrow-modulus path (p < 2^31: one signed 32-bit limb per residue, R = 2^32); see narrow_gen.inc.hpp.
// MatArg then holds int32 constants (896 entries): centred representatives of M_ji * 2^32 mod p, rows back to back.
struct N31Params {
    uint32_t p;        // modulus < 2^31
    uint32_t pinv;     // -p^-1 mod 2^32
    uint32_t h;        // (p + 1) / 2: v >= h is centred to v - p
    uint32_t pad;
};

// The library reports what it ran (include/sda_hip_debug.h: sda_debug_last_kernel): every share-generation launcher names
// the kernel instance it is about to launch, as rocprofv3 will print it (sda_capi.cpp keeps the last name per thread).
void note_kernel(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// strides in elements
struct GenLayout {
    const int64_t* secrets;   size_t secrets_stride;
    const int64_t* rand;      size_t rand_stride;     // rand == nullptr -> DRBG
    int64_t* out;             size_t out_stride_participant; size_t out_stride_clerk;
    size_t participants;      size_t len;             // secrets per participant
    uint64_t first_participant;                       // DRBG stream id of participant 0
    // packed Shamir with the device CSPRNG (rand == nullptr), "systematic" share map: output rows 0 .. direct_rows-1 ARE
    // the t draws of each batch (direct_rows = t) and the matrix the kernel is handed has the other n - direct_rows rows
    // (the polynomial through (1, 0), the secrets at omega_secrets^1..k and the draws at omega_shares^1..t).  0: every
    // row is a dot product with the n-row tss matrix (draws are the values at omega_secrets^(k+1..k+t)).
    uint32_t direct_rows;
};

// ---- share generation ---------------------------------------------------------------------------
// additive: k = 1, n shares, n-1 randoms per element (additive.rs:32-51 through batched.rs:18-53)
hipError_t launch_additive_generate(const GenLayout& L, uint32_t n, const ModParams& mod,
                                    const DrbgKey& key, int rounds, hipStream_t s);

// packed Shamir, fast path: (k, t) must be one of the compiled pairs and n*(k+t) <= SDA_MAT_ARG_MAX
bool packed_fast_path_available(uint32_t k, uint32_t t, uint32_t n);
hipError_t launch_packed_generate(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t,
                                  const ModParams& mod, const MontParams& mont, const MatArg& Mmont,
                                  const DrbgKey& key, int rounds, hipStream_t s);

// packed Shamir, compiled (k, t) shapes: balanced 31-bit limbs, carry-free v_mad_i64_i32 dot products
bool packed_l31_path_available(uint32_t k, uint32_t t, uint32_t n);
unsigned packed_l31_r_bits(uint32_t k, uint32_t t);     // 62 or 93: the Montgomery radix 2^bits the kernels of this shape expect
unsigned packed_l31_rt_r_bits(uint32_t kt);             // ... and the run-time (k, t) kernels (the global-matrix form serves compiled shapes too when n is large)
bool packed_l31_three_digit_compiled(uint32_t k, uint32_t t);   // a COMPILED (k, t) instance in the three-digit form
hipError_t launch_packed_generate_l31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t,
                                      const ModParams& mod, const L31Params& lp, const MatArg& Ml31,
                                      const DrbgKey& key, int rounds, hipStream_t s);

// packed Shamir over a NARROW prime (p < 2^31, the reference's own valid domain): run-time (k, t), k + t <= 16, ChaCha20 only
bool packed_n31_path_available(uint32_t k, uint32_t t, uint32_t rows, uint64_t p);
hipError_t launch_packed_generate_n31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                      const N31Params& np, const MatArg& M, const DrbgKey& key, hipStream_t s);
hipError_t launch_fused_packed_n31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod, const N31Params& np,
                                   const MatArg& M, const DrbgKey& key, uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev,
                                   size_t prev_rows, size_t jobs, size_t dimension, hipStream_t s, bool* fused);

// packed Shamir as a limb GEMM on the matrix cores (v_mfma_i32_16x16x64_i8 on balanced base-256 digits), compiled shapes
bool packed_mfma_path_available(uint32_t k, uint32_t t, uint32_t n);
hipError_t launch_packed_generate_mfma(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                       const MontParams& mont, const uint64_t* d_Mbal /* [n][8 ceil((k+t)/8)] */,
                                       const DrbgKey& key, int rounds, hipStream_t s);

hipError_t launch_fused_packed_mfma(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                    const MontParams& mont, const uint64_t* d_Mbal, const DrbgKey& key, int rounds, uint64_t* acc_lo,
                                    int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs, size_t dimension,
                                    hipStream_t s, bool* fused);

// the limb-31 kernel with run-time (k, t) and the matrix in global memory: any n, k + t <= 64
bool packed_l31_global_path_available(uint32_t k, uint32_t t);
hipError_t launch_packed_generate_l31_global(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                             const L31Params& lp, const uint64_t* d_Ml31 /* n (k+t) entries + 7 zeros */,
                                             const DrbgKey& key, int rounds, hipStream_t s);

// packed Shamir, transform form (tss's own radix-2 inverse / radix-3 forward structure) for large tss-valid shapes:
// k + t + 1 = 2^a = ord(omega_secrets), n + 1 = 3^b = ord(omega_shares), b >= 2, p < 2^62.  Every constant w travels with
// its companion floor(w 2^64 / p) (Shoup): table entries are (w, companion) pairs of 16 bytes.
struct FftPlan {
    uint32_t k, t, n;
    uint32_t m2, a;            // k + t + 1 = 2^a
    uint32_t m3, b;            // n + 1 = 3^b
    uint32_t G, lgG;           // batches per workgroup (a power of two): 16 or 8 (one CSPRNG block per draw serves 8) or 1
    uint32_t tw_lds;           // 1: the workgroup copies both twiddle tables to LDS
    uint64_t groups_padded;    // set by the launcher for G < 8: groups per participant rounded up to 8 * (16 / G) (XCD-aware group map)
    uint32_t lazy;             // 1 (narrow only): (4 b + 4) p < 2^32 - the radix-3 levels run without conditional subtractions
    uint64_t one_s;            // floor(2^32 / p): the companion of the constant 1 (narrow: full reduction of a lazy value)
    uint32_t narrow;           // 1: p < 2^30 - uint32_t values, tables of (uint32 w, uint32 floor(w 2^32 / p)) pairs, 32-bit companions below
    uint32_t nz_mask;          // bit 3 e0 + e1: e1 (m3 / 9) + e0 (m3 / 3) < m2, i.e. some 9-block of the zero-extended
                               // vector holds a coefficient at that position (fft_kernels.hip, the folded first two levels)
    const uint64_t* tw2;       // m2 / 2 pairs: omega_secrets^-j              (device)
    const uint64_t* tw3;       // m3 pairs:     omega_shares^j                (device)
    uint64_t omega, omega_s;   // omega_shares^(m3 / 3) and its companion
    uint64_t scale, scale_s;   // 1 / m2 and its companion
    uint32_t magic_k1, magic_t; // floor(2^32 / d) + 1 for d = k + 1 and d = t (exact quotients of the loader's small indices)
    uint32_t want_threads;     // host only, A/B: threads per workgroup (0 = the launcher's choice), knob SDA_FFT_THREADS at handle creation
    uint32_t no_xcd_map;       // host only, A/B: plain group order for G < 8 (knob SDA_NO_XCD_MAP at handle creation)
};
size_t fft_lds_bytes(uint32_t m2, uint32_t m3, uint32_t G, bool tw_lds, bool narrow);
hipError_t launch_packed_generate_fft(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const FftPlan& F, int rounds,
                                      hipStream_t s);

// packed Shamir over a narrow prime (p <= 0x7F7F7F, just below 2^23: three balanced base-256 digits per residue) as a limb GEMM on the
// matrix cores: any (k, t) with k + t <= 512, any n
// (ngemm_kernels.hip).  Draws are the transform kernel's (tss's nodes), ChaCha20 only.
struct NGemmPlan {
    uint32_t k, t, n;
    uint32_t ks;               // 64-term steps of the compiled instance: 1, 2, 4 or 8 (>= ceil((k + t) / 64), zero padded)
    uint32_t row_tiles;        // ceil(n / 16)
    int32_t c[5];              // centred representatives of 256^j 2^32 mod p: column j -> Montgomery operand
    N31Params np;
    const uint8_t* A;          // device: [row_tiles][ks][3 digits][64 lanes][16 bytes] - the A fragments of M's balanced digits,
                               // zero beyond n rows / k + t terms
};
size_t ngemm_tile_bytes(uint32_t ks);
bool packed_ngemm_path_available(uint32_t k, uint32_t t, uint64_t p);
uint32_t packed_ngemm_steps(uint32_t k, uint32_t t);
hipError_t launch_packed_generate_ngemm(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, hipStream_t s);
// dual-role form.  d_progress != nullptr (room for ngemm_clerk_slots(L, P) x 2 u64): the clerk sum of the previous tile rides in three
// clerk WAVES of every share-generation workgroup, a follow-up kernel sums what they did not get to (round 6, the default);
// d_progress == nullptr: clerk WORKGROUPS at fixed positions of the same grid (rounds 4 - 5; A/B knob SDA_NGEMM_CLERK_WG)
uint64_t ngemm_clerk_slots(const GenLayout& L, const NGemmPlan& P);
hipError_t launch_fused_packed_ngemm(const GenLayout& L, const ModParams& mod, const DrbgKey& key, const NGemmPlan& P, uint64_t* acc_lo,
                                     int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows, size_t jobs, size_t dimension, hipStream_t s,
                                     bool* fused, uint64_t* d_progress = nullptr, size_t progress_slots = 0);

// packed Shamir, any shape: matrix in global memory, randomness must be materialised (L.rand != 0)
hipError_t launch_packed_generate_generic(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t,
                                          const ModParams& mod, const MontParams& mont,
                                          const uint64_t* d_Mmont, hipStream_t s);

// materialise the DRBG draws exactly as the fused kernels would make them:
// out[p*stride + b*T + i], b < batches, i < T
hipError_t launch_drbg_fill(int64_t* d_out, size_t stride, size_t participants, size_t batches,
                            uint32_t T, uint64_t first_participant, const ModParams& mod,
                            const DrbgKey& key, int rounds, hipStream_t s);

// ---- clerk combine (combiner.rs:15-29) -------------------------------------------------------------
// acc (lo, hi) += sum over rows; element (job, row, i) at shares + job*job_stride + row*row_stride + i
hipError_t launch_combine_update(uint64_t* d_acc_lo, int64_t* d_acc_hi, const int64_t* d_shares,
                                 size_t jobs, size_t job_stride, size_t n_rows, size_t row_stride,
                                 size_t dimension, hipStream_t s, unsigned max_wg_per_cu = 0, unsigned walk_workgroups = 0);
hipError_t launch_combine_finish(const uint64_t* d_acc_lo, const int64_t* d_acc_hi, size_t count,
                                 const ModParams& mod, int64_t* d_out, hipStream_t s);

// ---- the reference's own signed representatives (value mode SDA_VALUES_RUST_SIGNED) - signed_kernels.hip --------------
// additive.rs:42-47: shares j < n-1 are the draws, untouched - injected (L.rand: (n-1) draws per element, ANY i64) or, with
// L.rand == nullptr, the sda-drbg-v1 draws of the canonical kernel made inside the kernel (no scratch)
hipError_t launch_additive_generate_signed(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                           hipStream_t s);
hipError_t launch_additive_generate_signed_drbg(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                                hipStream_t s);       // sda_kernels.hip (the DPP-quad CSPRNG lives there)
// combiner.rs:20-26: state[job][col] = (state + row) % q, rows in order; state is int64 [jobs][dimension]
hipError_t launch_combine_update_signed(int64_t* d_state, const int64_t* d_shares, size_t jobs, size_t job_stride, size_t n_rows,
                                        size_t row_stride, size_t dimension, int64_t q, hipStream_t s);
// (a + b) % q / (a - b) % q with Rust's truncated remainder, sums formed exactly
hipError_t launch_addsub_signed(const int64_t* d_a, const int64_t* d_b, size_t len, bool subtract, int64_t q, int64_t* d_out,
                                hipStream_t s);

// ---- dual-role launch: share generation of one tile + clerk-sum of the previous tile in ONE grid ----------
// d_prev holds the previous tile's shares in the SAME layout as L.out (job stride = L.out_stride_clerk, row
// stride = L.out_stride_participant), prev_rows participants; L.participants may be 0 (clerk-sum only) and
// d_prev may be null (generation only).  *fused = false (and nothing launched) when the layout or the shape
// rules the fused form out - the caller then issues the two ordinary launches.
hipError_t launch_fused_packed_l31(const GenLayout& L, uint32_t n, uint32_t k, uint32_t t, const ModParams& mod,
                                   const L31Params& lp, const MatArg& M, const DrbgKey& key, int rounds,
                                   uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows,
                                   size_t jobs, size_t dimension, hipStream_t s, bool* fused);
hipError_t launch_fused_additive(const GenLayout& L, uint32_t n, const ModParams& mod, const DrbgKey& key, int rounds,
                                 uint64_t* acc_lo, int64_t* acc_hi, const int64_t* d_prev, size_t prev_rows,
                                 size_t jobs, size_t dimension, hipStream_t s, bool* fused);

// ---- packed reconstruct (batched.rs:68-97 + tss reconstruct as a k x n' Lagrange matrix) ------------
hipError_t launch_packed_reconstruct(const int64_t* d_shares, size_t row_stride, uint32_t n_rows,
                                     uint32_t k, size_t batches, size_t dimension,
                                     const ModParams& mod, const MontParams& mont,
                                     const uint64_t* d_Rmont /*[k][n_rows]*/, int64_t* d_out,
                                     hipStream_t s);

// the same over a narrow prime (p < 2^31): one-limb arithmetic, R31 = centred Montgomery-form (R = 2^32) int32 constants [k][n_rows]
bool packed_reconstruct_n31_available(uint32_t n_rows, uint32_t k, uint64_t p, const int64_t* d_shares, size_t row_stride,
                                      const int64_t* d_out);
hipError_t launch_packed_reconstruct_n31(const int64_t* d_shares, size_t row_stride, uint32_t n_rows, uint32_t k, size_t batches,
                                         size_t dimension, const ModParams& mod, const N31Params& np, const int32_t* d_R31,
                                         int64_t* d_out, hipStream_t s);

// ---- element-wise masking (full.rs / chacha.rs mask+unmask arithmetic) ----------------------------
// out = (a + b) mod m   or   (a - b) mod m, any i64 inputs
hipError_t launch_addsub_mod(const int64_t* d_a, const int64_t* d_b, size_t len, bool subtract,
                             const ModParams& mod, int64_t* d_out, hipStream_t s);
// full.rs:21-35 with the on-device CSPRNG, `participants` vectors at once: mask[p][i] uniform (DRBG stream
// first_stream + p), masked[p][i] = (s[p][i] + mask[p][i]) mod m
hipError_t launch_full_mask_drbg(const int64_t* d_secrets, size_t secrets_stride, size_t participants, size_t len,
                                 uint64_t first_stream, const ModParams& mod, const DrbgKey& key, int rounds,
                                 int64_t* d_mask, size_t mask_stride, int64_t* d_masked, size_t masked_stride, hipStream_t s);

// ---- rand-0.3 ChaChaRng mask expansion (chacha.rs:36-39, :60-73) ----------------------------------
// Adds the `dimension` masks of each of the n_seeds seeds into the 128-bit accumulators.
// d_seeds: n_seeds x 8 u32 key words (seed words beyond the given ones are 0).
// Fast path: candidate i of every seed is added at position i (correct unless the seed's stream
// contains a rejected candidate).  d_rejects[s] (zeroed by the caller) counts the rejected candidates of
// seed s among its first `dimension` and keeps the index of up to three of them; such seeds must then be
// corrected by launch_chacha_mask_shift (count <= 3) or launch_chacha_mask_slow(subtract_naive = true).
struct RejectRecord {
    uint32_t count;
    uint32_t pos[3];
};
hipError_t launch_chacha_mask_accumulate(const uint32_t* d_seeds, size_t n_seeds, size_t dimension,
                                         const ModParams& mod, uint64_t zone, uint64_t* d_acc_lo,
                                         int64_t* d_acc_hi, RejectRecord* d_rejects, hipStream_t s);
// Parallel correction of the seeds in d_list whose rejected candidates are all recorded (count 1..3): removing
// candidate x shifts every later mask by one, so position i >= x takes candidate i + (rejections so far) instead of
// candidate i - each lane recomputes two ChaCha blocks and adds the difference.  The last few positions reach
// past candidate `dimension`, which the fast kernel never tested: those lanes walk on until they have their mask.
hipError_t launch_chacha_mask_shift(const uint32_t* d_seeds, const uint32_t* d_list, size_t n_list,
                                    const RejectRecord* d_rejects, size_t dimension, const ModParams& mod,
                                    uint64_t zone, uint64_t* d_acc_lo, int64_t* d_acc_hi, hipStream_t s);
// exact sequential-order expansion (handles rejections) of the seeds named by d_list (n_list indices
// into d_seeds; d_list == nullptr -> seeds 0..n_list-1), added into the accumulators.  With
// subtract_naive the fast kernel's contribution for those seeds is taken back first.
hipError_t launch_chacha_mask_slow(const uint32_t* d_seeds, const uint32_t* d_list, size_t n_list,
                                   size_t dimension, const ModParams& mod, uint64_t zone,
                                   uint64_t* d_acc_lo, int64_t* d_acc_hi, bool subtract_naive,
                                   hipStream_t s);

// the same expansion applied to each participant's own vector: out[p][i] = (secrets[p][i] + mask_i(seed p)) mod m
// (chacha.rs:36-47 for a device-resident tile).  Fast pass for every participant (rejections recorded in d_rejects,
// zeroed by the caller), then the repair pass for the listed ones: shift list (1..3 rejections), exact-order list
// (more; d_exact_list == nullptr walks participants 0..n_exact-1 in stream order and needs no fast pass).
hipError_t launch_chacha_apply_fast(const uint32_t* d_seeds, size_t participants, size_t dimension, const ModParams& mod,
                                    uint64_t zone, const int64_t* d_secrets, size_t secrets_stride, int64_t* d_out,
                                    size_t out_stride, RejectRecord* d_rejects, hipStream_t s);
hipError_t launch_chacha_apply_repair(const uint32_t* d_seeds, const uint32_t* d_shift_list, size_t n_shift,
                                      const uint32_t* d_exact_list, size_t n_exact, const RejectRecord* d_rejects,
                                      size_t dimension, const ModParams& mod, uint64_t zone, const int64_t* d_secrets,
                                      size_t secrets_stride, int64_t* d_out, size_t out_stride, hipStream_t s);

// ---- zig-zag LEB128 codec of share vectors (sodium.rs:36-41, :83-89) - varint_kernels.hip -------------
struct VarintRows {
    const int64_t* values;    // row r at values + r*row_stride
    size_t rows, len, row_stride;
};
size_t varint_encode_blocks(size_t rows, size_t len);     // workgroups (= entries of the block-sum arrays)
size_t varint_decode_blocks(size_t n_bytes);
hipError_t launch_varint_lengths(const VarintRows& R, uint32_t* d_block_bytes, hipStream_t s);
size_t scan_aux_entries(size_t n);                        // u64 scratch entries launch_scan_u32 needs
hipError_t launch_scan_u32(const uint32_t* d_in, uint64_t* d_out, size_t n, uint64_t* d_total, uint64_t* d_aux,
                           hipStream_t s);
hipError_t launch_varint_write(const VarintRows& R, const uint64_t* d_block_off, uint8_t* d_out,
                               uint64_t* d_row_offsets, hipStream_t s);
hipError_t launch_varint_count(const uint8_t* d_bytes, size_t n_bytes, uint32_t* d_block_counts, hipStream_t s);
hipError_t launch_varint_decode(const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_block_val_off,
                                size_t rows, size_t len, size_t row_stride, int64_t* d_out, uint32_t* d_status,
                                hipStream_t s);
// where the encoded rows lie in the byte buffer: contiguous (row r = [offsets[r], offsets[r+1]), offsets == nullptr:
// one row = the whole buffer) or slotted (lengths != nullptr: row r = [r*slot, r*slot + lengths[r]))
struct RowRanges {
    const uint64_t* offsets;
    const uint64_t* lengths;
    uint64_t slot;
};
// single pass, one wave per row (rows are independent messages with known byte ranges); pays off with >= ~1000 rows
hipError_t launch_varint_stream_decode(const uint8_t* d_bytes, size_t n_bytes, const RowRanges& rr, size_t rows,
                                       size_t len, size_t row_stride, int64_t* d_out, uint32_t* d_status, hipStream_t s);
// single-pass encode into slots: row r -> d_out + r*slot_bytes (16-byte aligned), its length -> d_row_bytes[r]
hipError_t launch_varint_stream_encode(const VarintRows& R, uint8_t* d_out, size_t slot_bytes, uint64_t* d_row_bytes,
                                       hipStream_t s);
// wire format -> 128-bit clerk accumulators directly: rows = jobs x rows_per_job encoded vectors (job-major), 16 rows
// of a job per workgroup through a sliding LDS column window; acc layout [jobs][len]
hipError_t launch_varint_stream_combine(const uint8_t* d_bytes, size_t n_bytes, const RowRanges& rr, size_t jobs,
                                        size_t rows_per_job, size_t len, uint64_t* d_acc_lo, int64_t* d_acc_hi,
                                        uint32_t* d_status, hipStream_t s);
hipError_t launch_varint_rowcheck(const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_offsets, size_t rows,
                                  size_t len, const uint64_t* d_block_val_off, uint32_t* d_status, hipStream_t s);

// ---- RFC 4648 base64 of `Binary` payloads (protocol/src/helpers.rs:174-216) - wire_kernels.hip ----------------
// text row r at d_text + (d_offsets ? d_offsets[r] : r * text_slot), d_lengths[r] characters; raw row r at
// d_out + r * out_slot; max_chars / max_bytes bound the longest row (they size the grid)
hipError_t launch_base64_decode_rows(const uint8_t* d_text, const uint64_t* d_offsets, size_t text_slot,
                                     const uint64_t* d_lengths, size_t rows, size_t max_chars, uint8_t* d_out,
                                     size_t out_slot, uint64_t* d_out_bytes, uint32_t* d_status, uint32_t* d_row_status,
                                     hipStream_t s);
hipError_t launch_base64_encode_rows(const uint8_t* d_in, size_t in_slot, const uint64_t* d_in_bytes, size_t rows,
                                     size_t max_bytes, uint8_t* d_text, size_t text_slot, uint64_t* d_text_bytes,
                                     hipStream_t s);

// ---- libsodium sealed boxes, batched (sodium.rs:43, :78) - sealedbox_kernels.hip ----------------------------
// per-box state written by the setup kernel (first lane of the box's DPP quad) and read, wave-uniformly, by the bulk kernels
struct SboxState {
    uint32_t subkey[8];      // XSalsa20 subkey = HSalsa20(HSalsa20(X25519, 0), nonce[0:16])
    uint32_t n0, n1;         // nonce[16:24]
    uint32_t s[4];           // Poly1305 s
    uint32_t r64[5], rS[5];  // r^64 and r^1024, 26-bit limbs
    uint32_t bad, pad[7];    // 1: unusable box (shorter than 48 bytes, all-zero shared secret)
    uint32_t rpow[64][5];    // r^1 .. r^64
};
size_t sbox_regions(size_t max_msg_bytes);          // Poly1305 regions per box (sizes d_partial: rows * regions * 5 words)
// box r at d_boxes + r * slot (16-byte aligned), d_row_bytes[r] bytes; plaintext to d_out + r * out_slot (16-byte aligned)
hipError_t launch_sealedbox_open(const uint8_t pk[32], const uint8_t sk[32], const uint8_t* d_boxes, size_t slot,
                                 const uint64_t* d_row_bytes, size_t rows, size_t max_box_bytes, uint8_t* d_out, size_t out_slot,
                                 uint64_t* d_out_bytes, uint32_t* d_ok, uint32_t* d_status, SboxState* d_states,
                                 uint32_t* d_partial, hipStream_t s);
// message r at d_msgs + r * msg_slot sealed to d_pks[(r / rows_per_key) % n_pks] with the ephemeral secret d_esk[r]
hipError_t launch_sealedbox_seal(const uint8_t* d_esk, const uint8_t* d_pks, size_t n_pks, size_t rows_per_key,
                                 const uint8_t* d_msgs, size_t msg_slot, const uint64_t* d_msg_bytes, size_t rows,
                                 size_t max_msg_bytes, uint8_t* d_boxes, size_t slot, uint64_t* d_row_bytes, SboxState* d_states,
                                 uint32_t* d_partial, hipStream_t s);

// ---- misc ---------------------------------------------------------------------------------------
// out[i] = sum over g < parts of parts[g*part_stride + i]  mod m   (cross-GPU partial sums)
hipError_t launch_modsum_parts(const int64_t* d_parts, size_t parts, size_t part_stride, size_t len,
                               const ModParams& mod, int64_t* d_out, hipStream_t s);
hipError_t launch_fill_synthetic(int64_t* d_out, size_t participants, size_t len, size_t stride,
                                 uint64_t first_participant, uint64_t seed, const ModParams& mod,
                                 hipStream_t s);

}  // namespace sda
