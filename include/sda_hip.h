/*
 * sda_hip.h - C ABI of the MI355X (gfx950) secure-aggregation compute core.
 *
 * This is the drop-in boundary for the one hot path of snipsco/sda: the crate-private
 * `client::crypto` sharing / masking traits (reference: client/src/crypto/sharing/mod.rs:10-33,
 * client/src/crypto/masking/mod.rs:9-31).  Every entry point below replaces exactly one trait
 * method or factory of the reference and cites it.  The Rust-side binding a maintainer would add
 * is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All scalars are 64-bit signed integers, like the reference's `Secret = Mask = MaskedSecret =
 *    Share = i64` (client/src/crypto/mod.rs:33-36).  There is no floating point anywhere.
 *  - Value domain: inputs may be ANY i64 (the reference does not range-check);
 *    they are canonicalised on load.  Outputs are canonical residues in [0, modulus).  The
 *    reference's intermediates live in (-q, q) (Rust truncated `%`); equality with the reference is
 *    therefore modulo q for intermediates and bit-for-bit after `RecipientOutput::positive`
 *    (client/src/receive.rs:13-21), which is the identity on canonical values.
 *  - Every function returns an `int` status: 0 = ok, negative = error (enum sda_status).  Nothing
 *    throws or aborts across the ABI.  The message of the last failure on the calling thread is
 *    available from sda_last_error().
 *  - Ownership: the caller allocates every input and output buffer; the library owns only the
 *    handle (device scratch, stream, precomputed matrices).  No pointer is retained after return.
 *  - Handles are NOT thread-safe (the reference traits carry no Send/Sync bound); use one per thread.
 *  - All host-buffer calls are blocking.  The *_dev calls take device pointers and a hipStream_t
 *    (passed as void*; NULL = the device's default stream, which every handle uses for its own
 *    work) and are asynchronous on that stream.  Work issued on different streams is NOT ordered
 *    by the library - with ONE exception: a share combiner's accumulators.  Every *_dev call that reads or
 *    writes them (begin_dev, update_dev, the wire-fed updates, generate_combine_dev, finish_dev) first makes its
 *    stream wait for the previous such call when that one ran on another stream, so e.g. generate_combine_dev
 *    on stream A followed by finish_dev on stream B returns the complete sums (the caller still orders its
 *    own input and output BUFFERS across streams).
 *  - There is NO CPU fallback: every compute entry point returns SDA_ERR_NO_DEVICE when no gfx950
 *    device is usable.
 *  - Randomness: the reference draws from OsRng inside generate()/mask() (additive.rs:42-44,
 *    full.rs:24-26, chacha.rs:29-33, tss::share).  Here it is injectable: pass `rand` to reproduce a
 *    given draw sequence (parity tests), or NULL to use the handle's on-device CSPRNG (ChaCha20
 *    keyed from OS entropy; spec in DESIGN.md "sda-drbg-v1").
 *  - CSPRNG uniqueness contract: a handle's master key is 32 bytes of OS entropy, and EVERY call that
 *    draws from the CSPRNG runs under its own call key = KDF(master key, call index) (one ChaCha20 block
 *    on the host).  `first_participant` only offsets the streams INSIDE one call (participant p of the
 *    call uses stream first_participant + p), so repeating it in a later call, or mixing host and device
 *    calls on one handle, can never repeat a keystream.  Stream ids are 56 bits: first_participant +
 *    participants > 2^56 is refused (SDA_ERR_INVALID_ARGUMENT).
 *    Deterministic mode - TESTS AND BENCHMARKS ONLY: after sda_*_set_drbg_key(key) the given key is the
 *    stream key of every call and the caller's stream ids select the streams (host calls take 0, 1, 2, ...).
 *    Runs are then reproducible, and two calls with overlapping stream ids DO repeat their randomness:
 *    uniqueness is the caller's job in that mode, which is why no production caller should enter it.
 */
#ifndef SDA_HIP_H
#define SDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDA_HIP_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------------------------- */
enum sda_status {
    SDA_OK = 0,
    /* mirrors of the reference's error strings (SdaClientResult `Err("...")?`) */
    SDA_ERR_BATCH_INPUT_WRONG_LENGTH = -1,   /* "Batch input wrong length"            additive.rs:33      */
    SDA_ERR_SHARING_FAILED = -2,             /* "Sharing failed for packed secret sharing scheme" packed_shamir.rs:41 */
    SDA_ERR_INPUTS_MUST_HAVE_SAME_LENGTH = -3, /* "Inputs must have same length"      packed_shamir.rs:74 */
    SDA_ERR_NOT_ENOUGH_SHARES = -4,          /* "Not enough shares to reconstruct"    packed_shamir.rs:75 */
    SDA_ERR_WRONG_DIMENSION = -5,            /* "Wrong dimension"                     combiner.rs:21      */
    SDA_ERR_MISMATCHING_DIMENSION = -6,      /* "Mismatching dimension"               additive.rs:64      */
    /* the masking traits are infallible and panic via assert!/assert_eq!
     * (chacha.rs:26,83; full.rs:43,58; none.rs:23,30); a Rust shim maps this code to panic!() */
    SDA_ERR_ASSERTION = -7,
    /* conditions with no counterpart in the reference */
    SDA_ERR_INVALID_ARGUMENT = -8,           /* NULL pointer, bad scheme parameters, short buffer  */
    SDA_ERR_UNSUPPORTED = -9,                /* parameters outside what the kernels implement      */
    SDA_ERR_NO_DEVICE = -10,                 /* no usable gfx950 device - there is no CPU fallback */
    SDA_ERR_HIP = -11,                       /* a HIP runtime call or kernel launch failed         */
    SDA_ERR_ALLOC = -12,                     /* host or device allocation failed                   */
    SDA_ERR_STATE = -13,                     /* streaming call out of order (update before begin)  */
    SDA_ERR_ENTROPY = -14,                   /* getrandom() failed: no key material, nothing was generated */
    SDA_ERR_COMM = -15,                      /* an RCCL call failed (multi-GPU reduce)              */
    SDA_ERR_SODIUM_DECRYPTION = -16          /* "Sodium decryption failure"           encryption/sodium.rs:80 */
};

/* ---- scheme parameters (the wire enums stay intact) ---------------------------------------- */

/* LinearSecretSharingScheme - protocol/src/crypto.rs:79-114 */
enum sda_sharing_kind {
    SDA_SHARING_ADDITIVE = 0,       /* Additive { share_count, modulus }                                   */
    SDA_SHARING_PACKED_SHAMIR = 1   /* PackedShamir { secret_count, share_count, privacy_threshold,
                                                      prime_modulus, omega_secrets, omega_shares }         */
};

typedef struct sda_sharing_scheme {
    int32_t  kind;               /* enum sda_sharing_kind                                         */
    uint64_t share_count;        /* both variants                                                 */
    int64_t  modulus;            /* Additive.modulus | PackedShamir.prime_modulus  (2 <= m < 2^62) */
    uint64_t secret_count;       /* PackedShamir only (Additive: ignored, input_size() == 1)      */
    uint64_t privacy_threshold;  /* PackedShamir only                                             */
    int64_t  omega_secrets;      /* PackedShamir only                                             */
    int64_t  omega_shares;       /* PackedShamir only                                             */
} sda_sharing_scheme_t;

/* LinearMaskingScheme - protocol/src/crypto.rs:43-75 */
enum sda_masking_kind {
    SDA_MASKING_NONE = 0,    /* None                                       */
    SDA_MASKING_FULL = 1,    /* Full { modulus }                           */
    SDA_MASKING_CHACHA = 2   /* ChaCha { modulus, dimension, seed_bitsize } */
};

typedef struct sda_masking_scheme {
    int32_t  kind;          /* enum sda_masking_kind */
    int64_t  modulus;       /* Full, ChaCha          */
    uint64_t dimension;     /* ChaCha                */
    uint64_t seed_bitsize;  /* ChaCha                */
} sda_masking_scheme_t;

/* derived sizes - protocol/src/crypto.rs:120-153 */
uint64_t sda_scheme_input_size(const sda_sharing_scheme_t* s);               /* :120-126 */
uint64_t sda_scheme_output_size(const sda_sharing_scheme_t* s);              /* :129-135 */
uint64_t sda_scheme_privacy_threshold(const sda_sharing_scheme_t* s);        /* :138-144 */
uint64_t sda_scheme_reconstruction_threshold(const sda_sharing_scheme_t* s); /* :147-153 */
int      sda_masking_has_mask(const sda_masking_scheme_t* s);                /* :67-74   */

/* ---- library / device -------------------------------------------------------------------- */
int         sda_abi_version(void);
const char* sda_version(void);
const char* sda_build_id(void);              /* first 16 hex digits of the sha256 over the sources, internal headers and
                                                the public headers this binary was compiled from ("unknown" for a hand build);
                                                __graft_entry__.source_digest() recomputes it from a tree */
const char* sda_kernel_id(void);             /* the same over the DEVICE code only (.hip files + internal headers): which kernels this
                                                binary carries - what a table of counter measurements is valid for (ABI 6) */
int         sda_device_count(void);          /* number of visible HIP devices, 0 if none      */
int         sda_set_device(int ordinal);     /* device used by handles created afterwards     */
int         sda_device_pci_bus_id(int ordinal, char* out, size_t cap);   /* "0000:05:00.0"; cap >= 16; identifies the
                                                 physical GPU (bench.py counts the distinct devices of the ranks) */
const char* sda_strerror(int status);
const char* sda_last_error(void);            /* thread-local; "" if none                      */

/* device memory helpers so that a host language needs no HIP binding of its own */
int sda_dev_malloc(void** d_ptr, size_t bytes);
int sda_dev_free(void* d_ptr);
int sda_dev_upload(void* d_dst, const void* h_src, size_t bytes);
int sda_dev_download(void* h_dst, const void* d_src, size_t bytes);
int sda_dev_memset(void* d_dst, int value, size_t bytes);
int sda_dev_synchronize(void);

/* ---- value representation -------------------------------------------------------------------------------
 * SDA_VALUES_CANONICAL (default): every output is the canonical residue in [0, q) - equal to the reference's value modulo
 * q, identical after RecipientOutput::positive() (receive.rs:13-21).  SDA_VALUES_RUST_SIGNED: the reference's OWN
 * representatives, bit for bit - Rust's `%` on i64 truncates, so its intermediates live in (-q, q) with history-dependent
 * signs: additive.rs:42-47 (the n-1 draws as they are, the last share the fold of (acc - r) % q),
 * combiner.rs:20-26 and additive.rs:62-69 (result = (result + v) % q, participant after participant), full.rs:30,46-48,62
 * and chacha.rs:41-44,88.  A fidelity mode for parity work and mixed deployments that compare intermediates: one lane
 * per column, participants strictly in order; it serves the trait-shaped calls, generate_batch_dev, the combiner's
 * begin / update[_dev] / finish[_dev], reconstruct[_dev] and unmask[_dev] (the dual-role launch, the wire-fed updates and
 * mask_batch_dev answer SDA_ERR_UNSUPPORTED in this mode).  Packed Shamir's generator / reconstructor refuse it
 * (SDA_ERR_UNSUPPORTED): their signed values are tss's, an un-vendored crate - compare those modulo the prime.  Set the
 * mode right after *_new (CHANGING a combiner's mode discards its running sums: begin again; setting the mode it already has
 * is a no-op).  generate_batch_dev in this mode without injected randomness makes its draws inside the kernel, from the
 * same sda-drbg-v1 streams as the canonical mode (the shares of the two modes are equal modulo q): no scratch buffer.
 *
 * Streams and the transform shapes.  For packed shapes served by the transform kernel (k + t > 32 over a prime above
 * 0x7F7F7F) generate_combine_dev issues the clerk sum of the previous tile on a low-priority side stream the GENERATOR owns,
 * joins it back into the `stream` of that call, and the combiner records that point: later calls on the combiner from ANY
 * stream are ordered after it by the library (see "Conventions").  sda_share_combiner_set_residency does not apply to the
 * side-stream sum (it runs a fixed grid). */
enum sda_value_mode { SDA_VALUES_CANONICAL = 0, SDA_VALUES_RUST_SIGNED = 1 };

/* ---- opaque handles: one per (scheme, role), like the reference's boxed trait objects ------ */
typedef struct sda_share_generator       sda_share_generator_t;
typedef struct sda_share_combiner        sda_share_combiner_t;
typedef struct sda_secret_reconstructor  sda_secret_reconstructor_t;
typedef struct sda_secret_masker         sda_secret_masker_t;
typedef struct sda_mask_combiner         sda_mask_combiner_t;
typedef struct sda_secret_unmasker       sda_secret_unmasker_t;

/* =============================================================================================
 * ShareGenerator  (sharing/mod.rs:10-17; impl batched.rs:18-53, additive.rs:32-51,
 *                  packed_shamir.rs:40-43 -> tss::packed::PackedSecretSharing::share)
 * ============================================================================================= */

/* new_share_generator(&scheme) - sharing/mod.rs:35-55 */
/* value representation of a handle's outputs (enum sda_value_mode above) */
int  sda_share_generator_set_value_mode(sda_share_generator_t* g, int mode);
int  sda_share_combiner_set_value_mode(sda_share_combiner_t* c, int mode);
int  sda_secret_reconstructor_set_value_mode(sda_secret_reconstructor_t* r, int mode);
int  sda_secret_masker_set_value_mode(sda_secret_masker_t* m, int mode);
int  sda_mask_combiner_set_value_mode(sda_mask_combiner_t* c, int mode);
int  sda_secret_unmasker_set_value_mode(sda_secret_unmasker_t* u, int mode);

int  sda_share_generator_new(const sda_sharing_scheme_t* scheme, sda_share_generator_t** out);
void sda_share_generator_free(sda_share_generator_t* g);

/* batch geometry: k = batch_input_size, n = batch_output_size, B = ceil(len / k)  (batched.rs:21-23) */
uint64_t sda_share_generator_share_count(const sda_share_generator_t* g);            /* n              */
uint64_t sda_share_generator_batch_count(const sda_share_generator_t* g, size_t len); /* B              */
uint64_t sda_share_generator_rand_count(const sda_share_generator_t* g, size_t len);  /* B * rand/batch */

/* TEST / BENCH ONLY - deterministic mode (see "CSPRNG uniqueness contract" above).
 *   set_drbg_key        : `key` becomes the stream key of every call; stream ids are the caller's
 *                         (device calls: first_participant + p; host calls: 0, 1, 2, ...).
 *   set_drbg_master_key : `key` replaces the OS-entropy master key but the per-call key derivation stays on
 *                         (call i runs under KDF(key, i)): pins the derivation itself in the parity tests.
 *   set_drbg_rounds     : ChaCha rounds of the CSPRNG, 20 (default) / 12 / 8, for the A/B measurements of
 *                         DESIGN.md; refused with SDA_ERR_STATE unless one of the two setters above was called
 *                         first, so a production handle always runs ChaCha20. */
int sda_share_generator_set_drbg_key(sda_share_generator_t* g, const uint8_t key[32]);
int sda_share_generator_set_drbg_master_key(sda_share_generator_t* g, const uint8_t key[32]);
int sda_share_generator_set_drbg_rounds(sda_share_generator_t* g, int rounds);
/* Which draw rule of sda-drbg-v1 serves `modulus` (ABI 6): SDA_DRBG_RULE_WORD = one 64-bit candidate word per draw (every
 * modulus above 0x7F7F7F; rounds 1 - 4 used it for all moduli), SDA_DRBG_RULE_PAIRED = one candidate word per TWO draws (moduli
 * up to 0x7F7F7F since library 0.5: the shares a given deterministic-mode key produces over such a modulus differ from those of
 * library 0.4 and earlier).  A deterministic-mode user who stores keys can detect the stream layout with this call.  Negative:
 * an enum sda_status (modulus out of range). */
#define SDA_DRBG_RULE_WORD 1
#define SDA_DRBG_RULE_PAIRED 2
int sda_drbg_draw_rule(int64_t modulus);

/* CSPRNG share map (packed Shamir, rand == NULL only; ABI 4).  tss draws the polynomial's t free parameters as its values
 * at omega_secrets^(k+1 .. k+t) (packed_shamir.rs:42 -> tss share: values = [0] ++ secrets ++ randomness).  A caller who
 * INJECTS randomness gets exactly that map.  When the library draws by itself, every matrix-form kernel uses the
 * SYSTEMATIC parametrisation of the same polynomial family instead: the t draws of a batch ARE its shares 0 .. t-1 (the
 * values at omega_shares^1 .. omega_shares^t) and shares t .. n-1 are the values, at omega_shares^(t+1 .. n), of the one
 * polynomial of degree <= t + k with f(1) = 0, f(omega_secrets^i) = secret_i (i = 1..k) and f(omega_shares^(j+1)) = draw_j
 * (j < t).  t + k + 1 distinct points fix such a polynomial, so for fixed secrets draws <-> tss randomness is a bijection
 * and uniform draws give the SAME joint distribution of the n shares; t of the n modular dot products per batch
 * disappear ((3,4,8): half of them).  SDA_SHARE_MAP_TSS_NODES is kept for the transform kernel (tss-valid shapes with
 * k + t > 32 over a prime above 0x7F7F7F (~2^23), where the draws are inputs of tss's own transform; up to that bound those shapes run
 * as a matrix product on the matrix cores and take the systematic map too - except under set_drbg_rounds(12 / 8), an A/B
 * setting the limb GEMM does not serve: those calls go through the transform kernel, hence tss's map, and
 * csprng_share_map() says so), for t = 0, when a share point collides with a node,
 * and on request (set_csprng_share_map: A/B measurements, round-3 fixtures).  Draw range: the device CSPRNG draws
 * uniformly from [0, p); tss 0.2 draws from [0, p - 1) (rand's Range::new(0, prime - 1)) - the library's range is the
 * one the secrecy argument wants (uniform over the field), and no reconstruction can tell the two apart. */
enum sda_share_map { SDA_SHARE_MAP_TSS_NODES = 0, SDA_SHARE_MAP_SYSTEMATIC = 1 };
int sda_share_generator_csprng_share_map(const sda_share_generator_t* g);          /* the map the NEXT rand == NULL call uses */
/* which kernel family the library selected for this scheme, "wide" or "wide+narrow" (sda_amd/csrc/path_select.hpp:
 * additive | generic mont64 l31 l31_global mfma fft, + n31 | ngemm over narrow primes) - reporting only; every family
 * computes the same shares (sharing/mod.rs:37-53 dispatches on the scheme enum alone) */
const char* sda_share_generator_path_name(const sda_share_generator_t* g);
int sda_share_generator_set_csprng_share_map(sda_share_generator_t* g, int map);   /* SDA_ERR_UNSUPPORTED where it does not exist */

/* generate(&mut self, secrets) -> Vec<Vec<Share>>   - sharing/mod.rs:14-17, batched.rs:18-53.
 *   secrets[len]               any i64
 *   rand[rand_len] or NULL     the draw sequence the reference would take from OsRng: for batch
 *                              b = 0..B-1, (n-1) values (additive.rs:42-44) or privacy_threshold
 *                              values (tss share), each taken mod modulus.  rand_len must equal
 *                              sda_share_generator_rand_count(g, len).
 *   out[n * B]                 clerk-major: out[j*B + b] = share of batch b for clerk j
 *                              (batched.rs:46-48).  The last batch is zero-padded (batched.rs:37-43). */
int sda_share_generator_generate(sda_share_generator_t* g,
                                 const int64_t* secrets, size_t len,
                                 const int64_t* rand, size_t rand_len,
                                 int64_t* out, size_t out_len);

/* P participants at once, everything resident in HBM (the bench / multi-GPU form).
 *   d_secrets : participant p at d_secrets + p*secrets_stride, `len` values each
 *   d_rand    : NULL -> on-device CSPRNG under this call's own key, stream id = first_participant + p
 *               (any value, e.g. 0, is safe in production - see the uniqueness contract above); else
 *               participant p's draws at d_rand + p*rand_stride (rand_count values each)
 *   d_out     : share (p, clerk j, batch b) at d_out + p*out_stride_participant
 *                                                   + j*out_stride_clerk + b
 *               e.g. job-major [n][P][B] (what the server's snapshot transposition produces,
 *               server/src/stores.rs:86-101): out_stride_clerk = P*B, out_stride_participant = B.
 *   All strides in elements.  Asynchronous on `stream`. */
int sda_share_generator_generate_batch_dev(sda_share_generator_t* g,
                                           const int64_t* d_secrets, size_t participants, size_t len,
                                           size_t secrets_stride,
                                           const int64_t* d_rand, size_t rand_stride,
                                           uint64_t first_participant,
                                           int64_t* d_out, size_t out_stride_participant,
                                           size_t out_stride_clerk,
                                           void* stream);

/* =============================================================================================
 * ShareCombiner  (sharing/mod.rs:19-25; impl combiner.rs:15-29)
 * ============================================================================================= */

/* new_share_combiner(&scheme) - sharing/mod.rs:57-73 */
int  sda_share_combiner_new(const sda_sharing_scheme_t* scheme, sda_share_combiner_t** out);
void sda_share_combiner_free(sda_share_combiner_t* c);

/* combine(&self, &Vec<Vec<Share>>) -> Vec<Share>   - combiner.rs:15-29.
 * rows[p] has row_lens[p] values.  dimension = row_lens[0] (0 rows -> *out_len = 0, combiner.rs:17);
 * a row of another length -> SDA_ERR_WRONG_DIMENSION (combiner.rs:21).  out_cap >= dimension. */
int sda_share_combiner_combine(sda_share_combiner_t* c,
                               const int64_t* const* rows, const size_t* row_lens, size_t n_rows,
                               int64_t* out, size_t out_cap, size_t* out_len);

/* same, for a dense [n_rows][dimension] matrix with row stride `row_stride` (elements) */
int sda_share_combiner_combine_dense(sda_share_combiner_t* c,
                                     const int64_t* shares, size_t n_rows, size_t dimension,
                                     size_t row_stride, int64_t* out);

/* Streaming / accumulating combiner: fixes the waste the reference acknowledges at
 * client/src/clerk.rs:71-72 (all P share vectors materialised before combining).  State is
 * `jobs` independent accumulator vectors of `dimension` exact 128-bit sums held in HBM.
 *   begin  : zero the state
 *   update : add n_rows rows to every job: element (job, row, i) at
 *            d_shares + job*job_stride + row*row_stride + i
 *   finish : reduce mod q into d_out[jobs * dimension] (canonical); state stays valid */
int sda_share_combiner_begin_dev(sda_share_combiner_t* c, size_t jobs, size_t dimension, void* stream);
int sda_share_combiner_update_dev(sda_share_combiner_t* c,
                                  const int64_t* d_shares, size_t job_stride,
                                  size_t n_rows, size_t row_stride, void* stream);
int sda_share_combiner_finish_dev(sda_share_combiner_t* c, int64_t* d_out, void* stream);

/* Scheduling knob for the device form: cap the clerk-sum kernel at `max_workgroups_per_cu` resident
 * workgroups per CU (0 = no cap).  The kernel is HBM-bound and 2 workgroups (8 waves) per CU already
 * saturate HBM; capping it lets a VALU-bound kernel on ANOTHER stream (share generation of the next
 * tile) keep the remaining wave slots, so the two overlap (measured +6 % on BASELINE config 3). */
int sda_share_combiner_set_residency(sda_share_combiner_t* c, unsigned max_workgroups_per_cu);

/* Software-pipelined step (the performance form of generate_batch_dev + combiner update_dev): ONE
 * dual-role launch generates tile i+1's shares into d_out while the clerk sums of tile i's shares
 * (d_prev, written by the previous call, SAME layout and strides as d_out) are accumulated into `c`.
 * Share generation is VALU-bound and the clerk sum HBM-bound, so interleaving their workgroups in one grid
 * overlaps the two; the shares are still materialised in HBM and read back.  `c` must have been begun with
 * jobs = share_count and dimension = batches.  participants == 0 -> clerk-sum only (last tile);
 * prev_participants == 0 -> generation only (first tile).  Randomness: the on-device CSPRNG, stream ids
 * first_participant + p - identical shares to sda_share_generator_generate_batch_dev.
 * For the limb GEMM shapes (a prime below 2^23, k + t > 16) the clerk sum rides inside the share-generation kernel and a
 * short follow-up launch on the same stream finishes it; the two communicate through scratch memory the GENERATOR owns:
 * consecutive calls on one generator go on ONE stream (or are ordered by the caller), like every other use of a handle. */
int sda_share_generator_generate_combine_dev(sda_share_generator_t* g, sda_share_combiner_t* c,
                                             const int64_t* d_secrets, size_t participants, size_t len,
                                             size_t secrets_stride, uint64_t first_participant,
                                             int64_t* d_out, size_t out_stride_participant, size_t out_stride_clerk,
                                             const int64_t* d_prev, size_t prev_participants, void* stream);

/* host-buffer streaming form (tiles are uploaded, accumulated, discarded); ONE job: update/finish on a combiner begun
 * with begin_dev(jobs != 1) are refused with SDA_ERR_STATE */
int sda_share_combiner_begin(sda_share_combiner_t* c, size_t dimension);
int sda_share_combiner_update(sda_share_combiner_t* c, const int64_t* shares, size_t n_rows,
                              size_t row_stride);
int sda_share_combiner_finish(sda_share_combiner_t* c, int64_t* out);

/* =============================================================================================
 * SecretReconstructor  (sharing/mod.rs:27-33; impl additive.rs:55-73, batched.rs:68-97,
 *                       packed_shamir.rs:73-77 -> tss ...::reconstruct)
 * ============================================================================================= */

/* new_secret_reconstructor(&scheme, dimension) - sharing/mod.rs:75-96 */
int  sda_secret_reconstructor_new(const sda_sharing_scheme_t* scheme, size_t dimension,
                                  sda_secret_reconstructor_t** out);
void sda_secret_reconstructor_free(sda_secret_reconstructor_t* r);

/* reconstruct(&self, &Vec<(usize, Vec<Share>)>) -> Vec<Secret>.
 *   indices[i]  clerk index of rows[i] (position in committee.clerks_and_keys, receive.rs:131-136)
 *   Additive   : column sum mod q; *out_len = row_lens[0]; ragged -> SDA_ERR_MISMATCHING_DIMENSION
 *                (additive.rs:55-73; indices ignored, the configured dimension too).
 *   PackedShamir: *out_len = dimension; n_rows < t+k -> SDA_ERR_NOT_ENOUGH_SHARES
 *                (packed_shamir.rs:75); every row needs >= ceil(dimension/k) values (batched.rs:84
 *                would panic on a short row: SDA_ERR_ASSERTION). */
int sda_secret_reconstructor_reconstruct(sda_secret_reconstructor_t* r,
                                         const size_t* indices,
                                         const int64_t* const* rows, const size_t* row_lens,
                                         size_t n_rows,
                                         int64_t* out, size_t out_cap, size_t* out_len);

/* device form: row i at d_shares + i*row_stride (row_len values each); d_out[*out_len] */
int sda_secret_reconstructor_reconstruct_dev(sda_secret_reconstructor_t* r,
                                             const size_t* indices, size_t n_rows,
                                             const int64_t* d_shares, size_t row_len, size_t row_stride,
                                             int64_t* d_out, size_t out_cap, size_t* out_len,
                                             void* stream);

/* =============================================================================================
 * SecretMasker / MaskCombiner / SecretUnmasker  (masking/mod.rs:9-31; impl none.rs, full.rs, chacha.rs)
 * ============================================================================================= */

/* new_secret_masker(&scheme) - masking/mod.rs:33-53 */
int  sda_secret_masker_new(const sda_masking_scheme_t* scheme, sda_secret_masker_t** out);
void sda_secret_masker_free(sda_secret_masker_t* m);
/* TEST / BENCH ONLY, as for the share generator */
int  sda_secret_masker_set_drbg_key(sda_secret_masker_t* m, const uint8_t key[32]);
int  sda_secret_masker_set_drbg_master_key(sda_secret_masker_t* m, const uint8_t key[32]);
int  sda_secret_masker_set_drbg_rounds(sda_secret_masker_t* m, int rounds);

/* length of the mask vector mask() returns for `len` secrets: None 0 (none.rs:15), Full len
 * (full.rs:24-26), ChaCha ceil(seed_bitsize/32) seed words (chacha.rs:31,48-50) */
uint64_t sda_secret_masker_mask_len(const sda_secret_masker_t* m, size_t len);

/* mask(&mut self, secrets) -> (Vec<Mask>, Vec<MaskedSecret>)  - none.rs:13-19, full.rs:21-35,
 * chacha.rs:24-54.
 *   rand: Full  -> `len` mask values (the OsRng draws of full.rs:24-26), or NULL for the CSPRNG
 *         ChaCha-> ceil(seed_bitsize/32) seed words, each used `as u32` (chacha.rs:30-33), or NULL
 *                  for OS entropy;   None -> ignored
 *   ChaCha with len != scheme.dimension -> SDA_ERR_ASSERTION (assert_eq!, chacha.rs:26) */
int sda_secret_masker_mask(sda_secret_masker_t* m,
                           const int64_t* secrets, size_t len,
                           const int64_t* rand, size_t rand_len,
                           int64_t* mask_out, size_t mask_cap, size_t* mask_len,
                           int64_t* masked_out);

/* device-resident batch: participate.rs:52-54 for a whole tile of participants.  Full: masks[p][i] uniform from the
 * on-device CSPRNG (stream first_participant + p; reproducible with sda_secret_masker_set_drbg_key),
 * masked[p][i] = (secrets[p][i] + masks[p][i]) mod q.  None: masked = secrets, d_masks untouched.  ChaCha: one
 * OS-entropy seed per participant (chacha.rs:29-33), its ceil(seed_bitsize/32) words written to d_masks[p][..]
 * (the "mask" a participant sends is its seed, chacha.rs:48-50; mask_stride >= that many), masked[p][i] =
 * (secrets[p][i] + i-th rand-0.3 ChaChaRng gen_range value of that seed) mod q; len must equal the scheme's dimension
 * (SDA_ERR_ASSERTION, chacha.rs:26); the call synchronises `stream`.  Full masks are combined on the recipient side
 * exactly like shares (full.rs:37-52 == combiner.rs:15-29): feed them to an sda_share_combiner begun with
 * jobs == 1; ChaCha seeds go to sda_mask_combiner_combine. */
int sda_secret_masker_mask_batch_dev(sda_secret_masker_t* m, const int64_t* d_secrets, size_t participants, size_t len,
                                     size_t secrets_stride, uint64_t first_participant, int64_t* d_masks,
                                     size_t mask_stride, int64_t* d_masked, size_t masked_stride, void* stream);

/* new_mask_combiner(&scheme) - masking/mod.rs:55-75 */
int  sda_mask_combiner_new(const sda_masking_scheme_t* scheme, sda_mask_combiner_t** out);
void sda_mask_combiner_free(sda_mask_combiner_t* c);

/* combine(&self, &Vec<Vec<Mask>>) -> Vec<Mask>
 *   None  : every row must be empty (assert!, none.rs:23); *out_len = 0
 *   Full  : column sum mod q (full.rs:37-52); ragged -> SDA_ERR_ASSERTION (assert_eq!, full.rs:43)
 *   ChaCha: rows are seeds (words used `as u32`, chacha.rs:62-64); every seed is re-expanded with
 *           rand-0.3 ChaChaRng + gen_range(0, modulus) and summed; *out_len = scheme.dimension
 *           (chacha.rs:56-77) */
int sda_mask_combiner_combine(sda_mask_combiner_t* c,
                              const int64_t* const* rows, const size_t* row_lens, size_t n_rows,
                              int64_t* out, size_t out_cap, size_t* out_len);

/* new_secret_unmasker(&scheme) - masking/mod.rs:77-94 */
int  sda_secret_unmasker_new(const sda_masking_scheme_t* scheme, sda_secret_unmasker_t** out);
void sda_secret_unmasker_free(sda_secret_unmasker_t* u);

/* unmask(&self, &(Vec<Mask>, Vec<MaskedSecret>)) -> Vec<Secret>   - none.rs:28-33, full.rs:54-67,
 * chacha.rs:79-93.   None: mask_len must be 0 (none.rs:30); Full/ChaCha: mask_len == masked_len
 * (assert_eq!, full.rs:58, chacha.rs:83) else SDA_ERR_ASSERTION.  out[masked_len]. */
int sda_secret_unmasker_unmask(sda_secret_unmasker_t* u,
                               const int64_t* mask, size_t mask_len,
                               const int64_t* masked, size_t masked_len,
                               int64_t* out);

/* the same on device-resident vectors (receive.rs:149-152 once the masks are combined): out = (masked - mask) mod q;
 * None: out = masked */
int sda_secret_unmasker_unmask_dev(sda_secret_unmasker_t* u, const int64_t* d_mask, const int64_t* d_masked, size_t len,
                                   int64_t* d_out, void* stream);

/* RecipientOutput::positive - client/src/receive.rs:13-21: v < 0 ? v + modulus : v.  Host-side,
 * element-wise; the identity on this library's canonical outputs. */
int sda_positive(const int64_t* values, size_t len, int64_t modulus, int64_t* out);

/* =============================================================================================
 * Share-vector wire codec (first "next" row): zig-zag LEB128 varints, the format either
 * side of the path - ShareEncryptor::encrypt encodes every share with `encode_var` before sealing
 * (client/src/crypto/encryption/sodium.rs:36-41) and ShareDecryptor::decrypt decodes until the reader
 * is empty (:83-89); integer-encoding 1.0 `VarInt for i64`.  The sealed box itself stays out of scope.
 * ============================================================================================= */
typedef struct sda_varint_codec sda_varint_codec_t;
int  sda_varint_codec_new(sda_varint_codec_t** out);
void sda_varint_codec_free(sda_varint_codec_t* c);
size_t sda_varint_max_encoded_size(size_t count);            /* 10 bytes per value */

/* encode `len` values -> out[*out_len] (sodium.rs:36-41) */
int sda_varint_encode(sda_varint_codec_t* c, const int64_t* values, size_t len,
                      uint8_t* out, size_t out_cap, size_t* out_len);
/* decode the whole byte string -> out[*out_len] values (sodium.rs:83-89).  A stream that ends inside a
 * value, or holds more than 10 bytes without a terminator, is refused with SDA_ERR_INVALID_ARGUMENT
 * (the reference would return garbage for it). */
int sda_varint_decode(sda_varint_codec_t* c, const uint8_t* bytes, size_t n_bytes,
                      int64_t* out, size_t out_cap, size_t* out_len);

/* device forms: `rows` vectors of `len` values (row r at d_values + r*row_stride) <-> one concatenated
 * byte stream; d_row_offsets[rows + 1] (device, u64) holds the byte offset of every row's encoding, so
 * each clerk's / participant's vector can be sealed or opened separately.
 *   encode: *total_bytes (host) is valid on return (the call synchronises once to size the output).
 *   decode: d_status (device u32) is OR-ed with 1 = malformed, 2 = a row does not hold `len` values,
 *           4 = a row ends inside a value; zero it beforehand.  d_row_offsets may be NULL for rows == 1. */
int sda_varint_encode_dev(sda_varint_codec_t* c, const int64_t* d_values, size_t rows, size_t len, size_t row_stride,
                          uint8_t* d_out, size_t out_cap, uint64_t* d_row_offsets, uint64_t* total_bytes, void* stream);
int sda_varint_decode_dev(sda_varint_codec_t* c, const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_row_offsets,
                          size_t rows, size_t len, int64_t* d_values, size_t row_stride, uint32_t* d_status, void* stream);

/* Slotted rows (single pass on both sides): every vector is sealed / opened on its own (sodium.rs:36-43, :78-89),
 * so the rows need not be contiguous.  Row r lives at d_bytes + r*slot_bytes (slot_bytes >= sda_varint_slot_size(len),
 * a multiple of 16; the buffer 16-byte aligned) and d_row_bytes[r] (device u64) is its encoded length.  Encoding
 * writes the lengths, decoding reads them; d_status as in sda_varint_decode_dev.  The bytes of a row are exactly
 * those of sda_varint_encode for that vector. */
size_t sda_varint_slot_size(size_t len);
int sda_varint_encode_rows_dev(sda_varint_codec_t* c, const int64_t* d_values, size_t rows, size_t len, size_t row_stride,
                               uint8_t* d_out, size_t slot_bytes, uint64_t* d_row_bytes, void* stream);
int sda_varint_decode_rows_dev(sda_varint_codec_t* c, const uint8_t* d_bytes, size_t slot_bytes, const uint64_t* d_row_bytes,
                               size_t rows, size_t len, int64_t* d_values, size_t row_stride, uint32_t* d_status, void* stream);

/* Streaming clerk (fixes the FIXME at client/src/clerk.rs:71-72): feed the opened
 * sealed-box payloads straight into the accumulating combiner - decode tile -> clerk-sum update ->
 * discard - instead of materialising all P decoded vectors (clerk.rs:80-86).
 *   _dev : `rows` encoded vectors in one device byte stream (row r = d_bytes[off[r] .. off[r+1])), each
 *          must decode to exactly the combiner's dimension; d_status as in sda_varint_decode_dev (if it
 *          comes back non-zero the running sums are invalid).  rows must be a multiple of the jobs the
 *          combiner was begun with, job-major: rows [j*rows/jobs, (j+1)*rows/jobs) belong to job j.  From
 *          1536 rows on nothing is materialised: the rows are streamed straight into the accumulators.
 *   host : ONE participant's encoded vector; a wrong value count -> SDA_ERR_WRONG_DIMENSION
 *          ("Wrong dimension", combiner.rs:21), a malformed stream -> SDA_ERR_INVALID_ARGUMENT.
 * Call sda_share_combiner_begin[_dev] first (the host form needs jobs == 1) and ..._finish[_dev] at the end. */
int sda_share_combiner_update_varint_dev(sda_share_combiner_t* c, sda_varint_codec_t* codec, const uint8_t* d_bytes,
                                         size_t n_bytes, const uint64_t* d_row_offsets, size_t rows,
                                         uint32_t* d_status, void* stream);
int sda_share_combiner_update_varint_rows_dev(sda_share_combiner_t* c, sda_varint_codec_t* codec, const uint8_t* d_bytes,
                                              size_t slot_bytes, const uint64_t* d_row_bytes, size_t rows,
                                              uint32_t* d_status, void* stream);   /* slotted rows, always streamed */
int sda_share_combiner_update_varint(sda_share_combiner_t* c, sda_varint_codec_t* codec, const uint8_t* bytes,
                                     size_t n_bytes);

/* =============================================================================================
 * Clerking-job container and `Binary` payloads.
 *
 * The reference hands a clerk its job as JSON: `ClerkingJob.encryptions: Vec<Encryption>` (protocol/src/resources.rs:
 * 128-139), every Encryption::Sodium(Binary) a base64 string (protocol/src/helpers.rs:174-216), assembled by the
 * server's snapshot transposition (server/src/stores.rs:86-101; server-store-mongodb/src/aggregations.rs:164-195).
 * "SDAJOBv1" is the same job as ONE contiguous binary blob that can be DMA'd into HBM and consumed in place:
 *
 *     offset  0  char  magic[8]        "SDAJOBv1"
 *             8  u32   header_bytes    64
 *            12  u32   payload_kind    enum sda_job_payload_kind
 *            16  u64   rows            number of encryptions (participants) in the job
 *            24  u64   slot_bytes      bytes reserved per row, a multiple of 16
 *            32  u64   lengths_offset  64: u64 row_bytes[rows], the byte length of every row's payload
 *            40  u64   payload_offset  first slot, 16-byte aligned: row r occupies [payload_offset + r * slot_bytes, + row_bytes[r])
 *            48  u64   total_bytes     payload_offset + rows * slot_bytes
 *            56  u64   reserved        0
 *     all integers little-endian.  With the blob at a 16-byte aligned device address `d`, (d + payload_offset,
 *     slot_bytes, d + lengths_offset) are exactly the (d_bytes, slot_bytes, d_row_bytes) arguments of the slotted-row
 *     device calls: sda_sealedbox_open_rows_dev (payload kind SEALED), sda_share_combiner_update_varint_rows_dev and
 *     sda_varint_decode_rows_dev (kind VARINT), sda_base64_decode_rows_dev (kind BASE64_TEXT).
 *
 * The container functions are host-only (no device needed).  INTEGRATION.md shows what stores.rs:86-101 would emit.
 * ============================================================================================= */
enum sda_job_payload_kind {
    SDA_JOB_SEALED = 0,       /* raw Encryption::Sodium bytes: sealed boxes (sodium.rs:43)                      */
    SDA_JOB_VARINT = 1,       /* opened payloads: zig-zag LEB128 share vectors (sodium.rs:36-41)                 */
    SDA_JOB_BASE64_TEXT = 2   /* the JSON form: base64 text of the sealed boxes (helpers.rs:174-216)             */
};
typedef struct sda_job_layout {
    uint32_t payload_kind;
    uint64_t rows, slot_bytes, lengths_offset, payload_offset, total_bytes;
} sda_job_layout_t;
size_t sda_job_slot_size(size_t max_payload_bytes);                   /* rounded up to 16                          */
size_t sda_job_container_size(size_t rows, size_t slot_bytes);        /* 0 if slot_bytes is not a multiple of 16   */
/* writes the header and a zeroed length table into buf[cap] */
int sda_job_container_init(uint8_t* buf, size_t cap, uint32_t payload_kind, size_t rows, size_t slot_bytes,
                           sda_job_layout_t* out /* may be NULL */);
/* cap = size of the caller's buffer: the header is re-validated against it before anything is written (the buffer may
 * be a parsed, untrusted blob) */
int sda_job_container_set_row(uint8_t* buf, size_t cap, size_t row, const uint8_t* payload, size_t len);
/* validates magic, geometry and every row length (a job is network input) */
int sda_job_container_parse(const uint8_t* buf, size_t n_bytes, sda_job_layout_t* out);
/* O(1): validates the header and THIS row's length (not the whole length table) */
int sda_job_container_get_row(const uint8_t* buf, size_t n_bytes, size_t row, const uint8_t** payload, size_t* len);

/* RFC 4648 base64 (standard alphabet, '=' padding, strict like data_encoding::base64::decode) of P payloads at once,
 * on the device.
 *   decode: text row r at d_text + (d_text_offsets ? d_text_offsets[r] : r * text_slot) - any alignment, e.g. straight
 *           into a JSON document - d_text_bytes[r] characters; raw bytes to d_out + r * out_slot (4-byte aligned rows),
 *           their count to d_out_bytes[r].  max_chars >= the longest row.  A malformed row (length not a multiple of
 *           4, foreign character, misplaced '=', stray bits under the padding: "Base64 decoding error",
 *           helpers.rs:183) ORs 8 into *d_status and sets d_row_status[r] (optional, zero it beforehand).  A row whose
 *           d_text_bytes[r] exceeds max_chars is malformed as well (the lengths are network input): d_out_bytes[r] = 0,
 *           the same status bits, and not one byte of it is read or written.
 *   encode: raw row r at d_in + r * in_slot (4-byte aligned rows) -> text row at d_text + r * text_slot (16-byte
 *           aligned rows, text_slot >= sda_base64_encoded_size(max_bytes)), its length to d_text_bytes[r].  A row
 *           with d_in_bytes[r] > max_bytes is refused: d_text_bytes[r] = 0, nothing read or written. */
size_t sda_base64_encoded_size(size_t n_bytes);
size_t sda_base64_decoded_max(size_t n_chars);
int sda_base64_decode_rows_dev(const uint8_t* d_text, const uint64_t* d_text_offsets, size_t text_slot,
                               const uint64_t* d_text_bytes, size_t rows, size_t max_chars, uint8_t* d_out, size_t out_slot,
                               uint64_t* d_out_bytes, uint32_t* d_status, uint32_t* d_row_status, void* stream);
int sda_base64_encode_rows_dev(const uint8_t* d_in, size_t in_slot, const uint64_t* d_in_bytes, size_t rows,
                               size_t max_bytes, uint8_t* d_text, size_t text_slot, uint64_t* d_text_bytes, void* stream);

/* =============================================================================================
 * Sealed boxes: libsodium crypto_box_seal / crypto_box_seal_open - X25519, XSalsa20-Poly1305,
 * nonce = BLAKE2b-24(epk || pk) - for a whole job of payloads at once, on the device.  The reference opens the P
 * encryptions of a clerking job one by one (clerk.rs:79-82 -> ShareDecryptor::decrypt, encryption/sodium.rs:72-92,
 * `sealedbox::open` at :78) and seals one payload per clerk in participate.rs:82-101 (`sealedbox::seal`, sodium.rs:43).
 * A sealed box is  epk(32) || tag(16) || ciphertext  = 48 bytes longer than its payload.
 *
 *   open_rows_dev : box r at d_boxes + r * slot_bytes, d_row_bytes[r] bytes (the SDAJOBv1 SEALED layout).  The
 *                   payload goes to d_out + r * out_slot and its length to d_out_bytes[r] - exactly the slotted rows
 *                   sda_share_combiner_update_varint_rows_dev consumes, so a job is opened, decoded and summed
 *                   without leaving HBM.  A box that does not authenticate (or is shorter than 48 bytes) gets length
 *                   0, d_ok[r] = 0 (optional array) and ORs 16 into *d_status: the reference fails the whole job with
 *                   "Sodium decryption failure" (sodium.rs:80) - check *d_status before using the sums.  The tags are
 *                   verified BEFORE the keystream pass and that pass skips the failing rows: the output slot of a box
 *                   that fails is left exactly as the caller passed it - no unauthenticated plaintext reaches d_out.
 *   seal_rows_dev : message r at d_msgs + r * msg_slot (d_msg_bytes[r] bytes) is sealed to
 *                   pks[(r / rows_per_key) % n_pks] (host array of n_pks 32-byte keys; job-major rows [n][P]:
 *                   rows_per_key = P; participant-major [P][n]: rows_per_key = 1) into d_boxes + r * slot_bytes, its
 *                   length to d_row_bytes[r].  esk: NULL = a fresh ephemeral key pair per box from OS entropy, as
 *                   crypto_box_seal draws it; else `rows` injected 32-byte ephemeral secrets (host) - TESTS ONLY, the
 *                   one way to compare a sealed box bit for bit.  A recipient key of small order (all-zero shared
 *                   secret - crypto_box_seal returns -1) is refused per row: d_row_bytes[r] = 0, the epk is written but
 *                   nothing is encrypted and no tag is stored; the host form returns SDA_ERR_INVALID_ARGUMENT and wipes
 *                   `out`.  A message longer than max_msg_bytes is refused the same way (length 0).
 *   A sealed-box handle serves ONE stream at a time: its device scratch (per-row key state, Poly1305 partials, staged
 *   keys) is shared by all calls, so two calls through one handle on different streams race - use one handle per stream.
 *   All rows and slots 16-byte aligned; max_* bound the longest row (they size the launch).  The host forms stage one
 *   payload through the device (ShareEncryptor::encrypt / ShareDecryptor::decrypt minus the varint codec).
 * ============================================================================================= */
#define SDA_SEALBYTES 48
typedef struct sda_sealedbox sda_sealedbox_t;
int  sda_sealedbox_new(sda_sealedbox_t** out);
void sda_sealedbox_free(sda_sealedbox_t* b);
int  sda_sealedbox_open_rows_dev(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t sk[32], const uint8_t* d_boxes,
                                 size_t slot_bytes, const uint64_t* d_row_bytes, size_t rows, size_t max_box_bytes,
                                 uint8_t* d_out, size_t out_slot, uint64_t* d_out_bytes, uint32_t* d_ok, uint32_t* d_status,
                                 void* stream);
int  sda_sealedbox_seal_rows_dev(sda_sealedbox_t* b, const uint8_t* pks, size_t n_pks, size_t rows_per_key, const uint8_t* esk,
                                 const uint8_t* d_msgs, size_t msg_slot, const uint64_t* d_msg_bytes, size_t rows,
                                 size_t max_msg_bytes, uint8_t* d_boxes, size_t slot_bytes, uint64_t* d_row_bytes, void* stream);
/* pk = X25519(sk, 9) (crypto_scalarmult_base), computed by the seal path's own ladder: lets tests and tools make key
 * pairs without libsodium; key generation proper stays with the reference's keystore */
int  sda_sealedbox_public_key(sda_sealedbox_t* b, const uint8_t sk[32], uint8_t pk[32]);
int  sda_sealedbox_seal(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t* esk /* NULL = OS entropy */,
                        const uint8_t* msg, size_t len, uint8_t* out, size_t out_cap);
int  sda_sealedbox_open(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t sk[32], const uint8_t* box, size_t len,
                        uint8_t* out, size_t out_cap, size_t* out_len);

/* =============================================================================================
 * Cross-GPU modular reduction (new; no reference counterpart - SURVEY.md 8e: the reference's parties meet over HTTP).
 * Participants are sharded across the GPUs of a node, one process per GPU, no collective on the data path; the
 * per-clerk partial sums meet ONCE at the end:
 *     direct reduce-scatter over the xGMI mesh (ncclSend/ncclRecv of 1/G slices, every GPU pair on its own link)
 *       -> exact modular sum of the G slices on the device -> direct all-gather of the reduced slices.
 * Never a sum collective on u64: 8 residues of a 62-bit modulus exceed 2^64.
 *
 *   sda_comm_unique_id : rank 0 makes the 128-byte RCCL id; the host hands it to the other ranks (any side channel)
 *   sda_comm_init      : collective over all ranks; binds the communicator to the calling thread's current device
 *                        (sda_set_device).  RCCL is loaded on first use (dlopen) - SDA_ERR_COMM if it is missing.
 *   sda_comm_rccl_version : the version code of the RCCL the library bound (ncclGetVersion: major*10000 + minor*100 + patch),
 *                        0 before the first sda_comm_unique_id / sda_comm_init or when RCCL has no such symbol.  Diagnostics only.
 *   sda_modular_allreduce_dev : d_partial[len] (any i64, this rank's partial sums) -> d_out[len] = sum over the ranks
 *                        mod modulus, canonical, on EVERY rank; asynchronous on `stream`; d_out may alias d_partial
 *                        only when world == 1.  len and modulus must agree on all ranks.
 *   sda_modsum_parts_dev : the local step on its own: d_out[len] = sum over `parts` vectors (part g at
 *                        d_parts + g*part_stride) mod modulus.
 * ============================================================================================= */
typedef struct sda_comm sda_comm_t;
#define SDA_COMM_ID_BYTES 128
int  sda_comm_unique_id(uint8_t id[SDA_COMM_ID_BYTES]);
int  sda_comm_init(const uint8_t id[SDA_COMM_ID_BYTES], int rank, int world, sda_comm_t** out);
void sda_comm_free(sda_comm_t* c);
int  sda_comm_rank(const sda_comm_t* c);
int  sda_comm_world(const sda_comm_t* c);
int  sda_comm_device(const sda_comm_t* c);   /* HIP ordinal the communicator was bound to at init; -1 for NULL */
int  sda_comm_rccl_version(void);
int  sda_modular_allreduce_dev(sda_comm_t* c, int64_t modulus, const int64_t* d_partial, size_t len, int64_t* d_out,
                               void* stream);
int  sda_modsum_parts_dev(int64_t modulus, const int64_t* d_parts, size_t parts, size_t part_stride,
                          size_t len, int64_t* d_out, void* stream);

/* Synthetic bench input: d_out[p*stride + i] = splitmix64(seed ^ ((first_participant+p) << 32 | i))
 * mod modulus.  Not on the product path. */
int sda_fill_synthetic_dev(int64_t* d_out, size_t participants, size_t len, size_t stride,
                           uint64_t first_participant, uint64_t seed, int64_t modulus, void* stream);

/* Timing hooks for bench.py: HIP events recorded on the stream the kernels are launched on
 * (torch.cuda.Event only sees torch's current stream). */
int sda_event_create(void** ev);
int sda_event_destroy(void* ev);
int sda_event_record(void* ev, void* stream);
int sda_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif
#endif /* SDA_HIP_H */
