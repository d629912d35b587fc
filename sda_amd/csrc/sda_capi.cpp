// C ABI of the MI355X secure-aggregation core (include/sda_hip.h).
//
// Host logic only: parameter validation mirroring the reference's error behaviour, one-time
// precomputation (Barrett / Montgomery constants, Lagrange matrices), staging of host buffers, and
// kernel launches.  All arithmetic on share/mask data happens in sda_kernels.hip on the GPU; there
// is no CPU fallback - without a usable gfx950 device every compute entry point fails with
// SDA_ERR_NO_DEVICE.
#include "../../include/sda_hip.h"

#include <hip/hip_runtime_api.h>
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <string.h>
#include <sys/random.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "capi_internal.hpp"
#include "host_chacha.hpp"
#include "host_math.hpp"
#include "kernels.hpp"
#include "modarith.hpp"
#include "path_select.hpp"

using namespace sda;

// -------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static thread_local int g_device = 0;

static int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(SDA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                               \
    } while (0)

#define SDA_TRY(expr)              \
    do {                           \
        int _s = (expr);           \
        if (_s != SDA_OK) return _s; \
    } while (0)

int capi_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

extern "C" const char* sda_strerror(int status) {
    switch (status) {
        case SDA_OK: return "ok";
        case SDA_ERR_BATCH_INPUT_WRONG_LENGTH: return "Batch input wrong length";
        case SDA_ERR_SHARING_FAILED: return "Sharing failed for packed secret sharing scheme";
        case SDA_ERR_INPUTS_MUST_HAVE_SAME_LENGTH: return "Inputs must have same length";
        case SDA_ERR_NOT_ENOUGH_SHARES: return "Not enough shares to reconstruct";
        case SDA_ERR_WRONG_DIMENSION: return "Wrong dimension";
        case SDA_ERR_MISMATCHING_DIMENSION: return "Mismatching dimension";
        case SDA_ERR_ASSERTION: return "assertion failed";
        case SDA_ERR_INVALID_ARGUMENT: return "invalid argument";
        case SDA_ERR_UNSUPPORTED: return "unsupported parameters";
        case SDA_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
        case SDA_ERR_HIP: return "HIP runtime error";
        case SDA_ERR_ALLOC: return "allocation failed";
        case SDA_ERR_STATE: return "call out of order";
        case SDA_ERR_ENTROPY: return "the operating system's entropy source failed";
        case SDA_ERR_COMM: return "RCCL communicator error";
        case SDA_ERR_SODIUM_DECRYPTION: return "Sodium decryption failure";
        default: return "unknown status";
    }
}

extern "C" const char* sda_last_error(void) { return g_last_error.c_str(); }
extern "C" int sda_abi_version(void) { return SDA_HIP_ABI_VERSION; }

// ---- path-selection knobs (capi_internal.hpp) --------------------------------------------------------------------------
// TWO builds of this translation unit (the only one that differs between them, __graft_entry__.build()):
//   libsda_hip.so       the release library: knob() is the constant 0, nothing can redirect kernel selection, no sda_debug_*
//                       entry point that changes or creates anything is exported;
//   libsda_hip_test.so  -DSDA_TEST_HOOKS: the knob table and the test-only entry points of include/sda_hip_debug.h.
#if defined(SDA_AB_KNOBS) && !defined(SDA_TEST_HOOKS)
#define SDA_TEST_HOOKS 1
#endif
#ifdef SDA_TEST_HOOKS
namespace {
const char* const kKnobNames[sda::KNOB_COUNT] = {
    "SDA_FORCE_GENERIC", "SDA_FORCE_MONT64", "SDA_FORCE_FFT", "SDA_FORCE_MFMA", "SDA_NO_MFMA", "SDA_NO_SIDE_STREAM",
    "SDA_SIDE_STREAM_WGS", "SDA_SIDE_STREAM_PRIORITY", "SDA_FFT_G", "SDA_FFT_THREADS", "SDA_VARINT_PATH", "SDA_FORCE_COLLECTIVES",
    "SDA_NO_NARROW", "SDA_WIRE_WG_PER_CU", "SDA_SBOX_WG_PER_CU", "SDA_NO_LAZY", "SDA_NO_XCD_MAP", "SDA_NO_NGEMM", "SDA_NO_WIDE_GROUP", "SDA_NGEMM_CLERK_WG", "SDA_NO_KARATSUBA"};
std::atomic<long> g_knobs[sda::KNOB_COUNT];
}  // namespace
long sda::knob(sda::Knob k) {
    const long v = g_knobs[k].load(std::memory_order_relaxed);
#ifdef SDA_AB_KNOBS
    if (v == 0)
        if (const char* e = getenv(kKnobNames[k])) {
            if (k == KNOB_VARINT_PATH) return !strcmp(e, "stream") ? 1 : !strcmp(e, "scan") ? 2 : 0;
            if (k == KNOB_SIDE_STREAM_PRIORITY_HIGH) return e[0] == 'h';
            const long n = atol(e);
            return n ? n : 1;                                   // "set" counts as 1 for the boolean knobs
        }
#endif
    return v;
}
extern "C" int sda_debug_set_knob(const char* name, long value) {
    if (!name) return fail(SDA_ERR_INVALID_ARGUMENT, "name is NULL");
    for (int k = 0; k < sda::KNOB_COUNT; ++k)
        if (!strcmp(name, kKnobNames[k])) { g_knobs[k].store(value, std::memory_order_relaxed); return SDA_OK; }
    return fail(SDA_ERR_INVALID_ARGUMENT, "unknown knob %s", name);
}
extern "C" void sda_debug_reset_knobs(void) {
    for (auto& k : g_knobs) k.store(0, std::memory_order_relaxed);
}
extern "C" int sda_debug_env_knobs_compiled_in(void) {
#ifdef SDA_AB_KNOBS
    return 1;
#else
    return 0;
#endif
}
extern "C" int sda_debug_hooks_compiled_in(void) { return 1; }
#else
long sda::knob(sda::Knob) { return 0; }                       // the release library: every knob is at its default, for good
extern "C" int sda_debug_hooks_compiled_in(void) { return 0; }
#endif
#ifdef SDA_TEST_HOOKS
extern "C" const char* sda_version(void) { return "sda-hip 0.6.0 (gfx950) +test-hooks"; }
#else
extern "C" const char* sda_version(void) { return "sda-hip 0.6.0 (gfx950)"; }
#endif

// ---- what ran, and what this binary was built from ------------------------------------------------------------------
// SDA_BUILD_ID is handed in by __graft_entry__.build(): the sha256 (first 16 hex digits) over the library's sources, internal
// headers and include/*.h.  smoke() and tests recompute it from the tree: a stale prebuilt .so cannot pass for a fresh one.
#ifndef SDA_BUILD_ID
#define SDA_BUILD_ID "unknown"
#endif
extern "C" const char* sda_build_id(void) { return SDA_BUILD_ID; }
#ifndef SDA_KERNEL_ID
#define SDA_KERNEL_ID "unknown"
#endif
extern "C" const char* sda_kernel_id(void) { return SDA_KERNEL_ID; }

static thread_local char g_last_gen_kernel[192];       // the last share-generation kernel instance launched on this thread
static thread_local char g_last_call_kernels[320];     // ... and the launches of the last generate call, as one string
void sda::note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_gen_kernel, sizeof g_last_gen_kernel, fmt, ap);
    va_end(ap);
}
extern "C" const char* sda_debug_last_kernel(void) { return g_last_call_kernels; }
#ifdef SDA_TEST_HOOKS
extern "C" int sda_debug_stream_create(void** stream) {
    if (!stream) return fail(SDA_ERR_INVALID_ARGUMENT, "stream is NULL");
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return SDA_OK;
}
extern "C" int sda_debug_stream_destroy(void* stream) {
    if (stream) HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return SDA_OK;
}
extern "C" int sda_debug_stream_synchronize(void* stream) {
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return SDA_OK;
}
extern "C" int sda_debug_mem_info(size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return SDA_OK;
}
#endif

// -------------------------------------------------------------------------------------------------
// device context
// -------------------------------------------------------------------------------------------------
extern "C" int sda_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int sda_set_device(int ordinal) {
    int n = sda_device_count();
    if (n == 0) return fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    if (ordinal < 0 || ordinal >= n) return fail(SDA_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (have %d)", ordinal, n);
    g_device = ordinal;
    HIP_TRY(hipSetDevice(ordinal));
    return SDA_OK;
}

extern "C" int sda_device_pci_bus_id(int ordinal, char* out, size_t cap) {
    if (!out || cap < 16) return fail(SDA_ERR_INVALID_ARGUMENT, "pci bus id needs a buffer of at least 16 bytes");
    out[0] = 0;
    int n = sda_device_count();
    if (n == 0) return fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    if (ordinal < 0 || ordinal >= n) return fail(SDA_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (have %d)", ordinal, n);
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)(cap > 64 ? 64 : cap), ordinal));
    return SDA_OK;
}

namespace {

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;

    int init() {
        int n = sda_device_count();
        if (n == 0) return fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
        device = g_device < n ? g_device : 0;
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SDA_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
        // All work is issued on the device's default (null) stream unless the caller passes one:
        // handles then serialise against each other (generate -> combine -> reconstruct) with no
        // cross-stream hazards.  Callers wanting overlap pass their own streams to the *_dev calls.
        stream = nullptr;
        return SDA_OK;
    }
    int use() const {
        HIP_TRY(hipSetDevice(device));
        return SDA_OK;
    }
    hipStream_t pick(void* user) const { return user ? reinterpret_cast<hipStream_t>(user) : stream; }
    int sync() const {
        HIP_TRY(hipStreamSynchronize(stream));
        return SDA_OK;
    }
    void destroy() { stream = nullptr; }
};

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SDA_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(SDA_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return SDA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    // for buffers that held secrets, shares, randomness or seeds: zero the memory before it returns to the allocator
    void wipe_release() {
        if (p) { (void)hipMemset(p, 0, cap); (void)hipDeviceSynchronize(); }
        release();
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

int make_mod(int64_t modulus, ModParams& mod) {
    if (modulus < 2) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus must be >= 2 (got %lld)", (long long)modulus);
    if ((uint64_t)modulus >= (1ull << 62))
        return fail(SDA_ERR_UNSUPPORTED, "modulus must be < 2^62 (got %lld): i64 add/sub of two residues must not overflow", (long long)modulus);
    mod.m = (uint64_t)modulus;
    mod.mu = h_barrett_mu(mod.m);
    mod.lemire_thr = h_lemire_thr(mod.m);
    mod.lemire_thr2 = h_lemire_thr2(mod.m);
    return SDA_OK;
}

}  // namespace
int capi_make_mod(int64_t modulus, sda::ModParams& mod) { return make_mod(modulus, mod); }
extern "C" int sda_drbg_draw_rule(int64_t modulus) {
    ModParams mod;
    if (int st = make_mod(modulus, mod)) return st;
    return drbg_paired(mod.m) ? SDA_DRBG_RULE_PAIRED : SDA_DRBG_RULE_WORD;
}
namespace {

int os_entropy(void* buf, size_t len) {
    uint8_t* b = static_cast<uint8_t*>(buf);
    size_t got = 0;
    while (got < len) {
        ssize_t r = getrandom(b + got, len - got, 0);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return fail(SDA_ERR_ENTROPY, "getrandom failed (errno %d)", errno);
        got += (size_t)r;
    }
    return SDA_OK;
}

void key_from_bytes(const uint8_t key[32], DrbgKey& k) {
    for (int i = 0; i < 8; ++i)
        k.w[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) |
                 ((uint32_t)key[4 * i + 3] << 24);
}

// sda-drbg-v1 key management.  Production: the master key is 32 bytes of OS entropy and EVERY call that
// draws from the CSPRNG runs under its own call key = KDF(master, call index) - the caller's `first_participant` only
// offsets the streams inside that call, so repeating it can never repeat a keystream.  Deterministic mode (tests,
// bench): set_drbg_key() makes the given key the stream key of every call and the caller's stream ids select the
// streams - reproducible, and therefore the caller's responsibility.
struct Drbg {
    DrbgKey key{};                  // master key (production) / stream key (deterministic mode)
    bool deterministic = false;
    bool derive = true;             // false only in deterministic mode
    int rounds = 20;                // 12 / 8 only through set_rounds() in deterministic mode (bench A/B)
    uint64_t next_stream = 0;       // host forms in deterministic mode
    uint64_t calls = 0;             // call-key index
    int init() {
        uint8_t raw[32];
        SDA_TRY(os_entropy(raw, sizeof raw));
        key_from_bytes(raw, key);
        explicit_bzero(raw, sizeof raw);
        return SDA_OK;
    }
    void set_stream_key(const uint8_t k[32]) { key_from_bytes(k, key); deterministic = true; derive = false; next_stream = 0; calls = 0; }
    void set_master_key(const uint8_t k[32]) { key_from_bytes(k, key); deterministic = true; derive = true; next_stream = 0; calls = 0; }
    int set_rounds(int r) {
        if (!deterministic) return fail(SDA_ERR_STATE, "the round count can only be changed on a handle in deterministic (test / bench) mode: call set_drbg_key first");
        if (r != 20 && r != 12 && r != 8) return fail(SDA_ERR_INVALID_ARGUMENT, "rounds must be 20, 12 or 8");
        rounds = r;
        return SDA_OK;
    }
    // key of one API call
    DrbgKey call_key() {
        if (!derive) return key;
        DrbgKey k;
        h_drbg_call_key(key.w, calls++, k.w);
        return k;
    }
    // stream id of a host-form call (one participant)
    uint64_t host_stream() { return derive ? 0 : next_stream++; }
    void wipe() { explicit_bzero(&key, sizeof key); }
};

// stream ids are 56 bits wide in the block layout (word 15 keeps 8 bits for the retry attempt)
int check_streams(uint64_t first, uint64_t count) {
    const uint64_t lim = 1ull << 56;
    if (first >= lim || count > lim - first)
        return fail(SDA_ERR_INVALID_ARGUMENT, "CSPRNG stream ids must stay below 2^56 (first_participant %llu + %llu participants)",
                    (unsigned long long)first, (unsigned long long)count);
    return SDA_OK;
}

bool sharing_is_additive(const sda_sharing_scheme_t* s) { return s->kind == SDA_SHARING_ADDITIVE; }

int check_scheme_kind(const sda_sharing_scheme_t* s) {
    if (!s) return fail(SDA_ERR_INVALID_ARGUMENT, "scheme is NULL");
    if (s->kind != SDA_SHARING_ADDITIVE && s->kind != SDA_SHARING_PACKED_SHAMIR)
        return fail(SDA_ERR_INVALID_ARGUMENT, "unknown sharing scheme kind %d", s->kind);
    return SDA_OK;
}

// 128-bit exact accumulator state living in HBM
struct AccState {
    DevBuf lo, hi;
    size_t count = 0;
    bool rust_signed = false;      // lo holds the reference's running values (int64, (-q, q)) instead of 128-bit sums
    int64_t q = 0;                 // ... and this is the modulus of their `%` (set with the mode)
    int reset(size_t n, hipStream_t s) {
        SDA_TRY(lo.reserve(n * 8));
        SDA_TRY(hi.reserve(n * 8));
        count = n;
        if (n) {
            HIP_TRY(hipMemsetAsync(lo.p, 0, n * 8, s));
            HIP_TRY(hipMemsetAsync(hi.p, 0, n * 8, s));
        }
        return SDA_OK;
    }
    void release() { lo.release(); hi.release(); count = 0; }
};

// one update / the final read-out of the running column sums, in either representation: exact 128-bit sums reduced once at
// the end (canonical), or the reference's own `result = (result + v) % q` participant after participant (rust_signed)
int acc_update(AccState& acc, const int64_t* d_shares, size_t jobs, size_t job_stride, size_t n_rows, size_t row_stride,
               size_t dimension, hipStream_t s, unsigned max_wg_per_cu = 0, unsigned walk = 0) {
    if (acc.rust_signed)
        HIP_TRY(launch_combine_update_signed(acc.lo.as<int64_t>(), d_shares, jobs, job_stride, n_rows, row_stride, dimension, acc.q, s));
    else
        HIP_TRY(launch_combine_update(acc.lo.as<uint64_t>(), acc.hi.as<int64_t>(), d_shares, jobs, job_stride, n_rows, row_stride,
                                      dimension, s, max_wg_per_cu, walk));
    return SDA_OK;
}
int acc_finish(const AccState& acc, size_t count, const ModParams& mod, int64_t* d_out, hipStream_t s) {
    if (acc.rust_signed) {
        if (count) HIP_TRY(hipMemcpyAsync(d_out, acc.lo.p, count * 8, hipMemcpyDeviceToDevice, s));
    } else {
        HIP_TRY(launch_combine_finish(acc.lo.as<uint64_t>(), acc.hi.as<int64_t>(), count, mod, d_out, s));
    }
    return SDA_OK;
}

// rows given as separate host vectors -> dense device tile uploads + combine_update
int accumulate_host_rows(const Ctx& ctx, AccState& acc, DevBuf& tile, const int64_t* const* rows, size_t n_rows,
                         size_t dimension) {
    if (n_rows == 0 || dimension == 0) return SDA_OK;
    const size_t tile_bytes_max = (size_t)256 << 20;
    size_t rows_per_tile = tile_bytes_max / (dimension * 8);
    if (rows_per_tile < 1) rows_per_tile = 1;
    if (rows_per_tile > n_rows) rows_per_tile = n_rows;
    const size_t stride = dimension + (dimension & 1);     // keep rows 16-byte aligned
    SDA_TRY(tile.reserve(rows_per_tile * stride * 8));
    for (size_t r0 = 0; r0 < n_rows; r0 += rows_per_tile) {
        const size_t nr = std::min(rows_per_tile, n_rows - r0);
        for (size_t r = 0; r < nr; ++r)
            HIP_TRY(hipMemcpyAsync(tile.as<int64_t>() + r * stride, rows[r0 + r], dimension * 8, hipMemcpyHostToDevice, ctx.stream));
        SDA_TRY(acc_update(acc, tile.as<int64_t>(), 1, 0, nr, stride, dimension, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));          // the tile is reused
    }
    return SDA_OK;
}

int column_sum_host(const Ctx& ctx, AccState& acc, DevBuf& tile, DevBuf& d_out, const ModParams& mod,
                    const int64_t* const* rows, size_t n_rows, size_t dimension, int64_t* out) {
    SDA_TRY(ctx.use());
    SDA_TRY(acc.reset(dimension, ctx.stream));
    SDA_TRY(accumulate_host_rows(ctx, acc, tile, rows, n_rows, dimension));
    SDA_TRY(d_out.reserve(dimension * 8));
    SDA_TRY(acc_finish(acc, dimension, mod, d_out.as<int64_t>(), ctx.stream));
    HIP_TRY(hipMemcpyAsync(out, d_out.p, dimension * 8, hipMemcpyDeviceToHost, ctx.stream));
    return ctx.sync();
}

// rand 0.3 Range<i64>: zone = u64::MAX - u64::MAX % range  
uint64_t rand03_zone(uint64_t range) { return UINT64_MAX - (UINT64_MAX % range); }

// Adds the masks of `n_seeds` ChaCha seeds (8 key words each, host) into acc.
int chacha_accumulate(const Ctx& ctx, const std::vector<uint32_t>& seeds8, size_t n_seeds, size_t dimension,
                      const ModParams& mod, AccState& acc, DevBuf& d_seeds, DevBuf& d_flags, DevBuf& d_list) {
    if (n_seeds == 0 || dimension == 0) return SDA_OK;
    const uint64_t zone = rand03_zone(mod.m);
    // expected rejected candidates per seed: up to about one, nearly every seed is either clean or repaired by the
    // parallel shift pass (<= 3 rejections); beyond that the exact-order kernel does everything
    const double p_rej = (double)(UINT64_MAX - zone + 1) / 18446744073709551616.0;
    const bool all_slow = p_rej * (double)dimension > 1.0 || dimension >= 0xFFFFFFF0ull;
    const size_t chunk = (size_t)1 << 20;                  // seeds per launch
    for (size_t s0 = 0; s0 < n_seeds; s0 += chunk) {
        const size_t ns = std::min(chunk, n_seeds - s0);
        SDA_TRY(d_seeds.reserve(ns * 32));
        HIP_TRY(hipMemcpyAsync(d_seeds.p, seeds8.data() + s0 * 8, ns * 32, hipMemcpyHostToDevice, ctx.stream));
        if (all_slow) {
            HIP_TRY(launch_chacha_mask_slow(d_seeds.as<uint32_t>(), nullptr, ns, dimension, mod, zone, acc.lo.as<uint64_t>(),
                                            acc.hi.as<int64_t>(), false, ctx.stream));
            HIP_TRY(hipStreamSynchronize(ctx.stream));
            continue;
        }
        SDA_TRY(d_flags.reserve(ns * sizeof(RejectRecord)));
        HIP_TRY(hipMemsetAsync(d_flags.p, 0, ns * sizeof(RejectRecord), ctx.stream));
        HIP_TRY(launch_chacha_mask_accumulate(d_seeds.as<uint32_t>(), ns, dimension, mod, zone, acc.lo.as<uint64_t>(),
                                              acc.hi.as<int64_t>(), d_flags.as<RejectRecord>(), ctx.stream));
        std::vector<RejectRecord> rec(ns);
        HIP_TRY(hipMemcpyAsync(rec.data(), d_flags.p, ns * sizeof(RejectRecord), hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        std::vector<uint32_t> shift, exact;                 // seeds repaired in parallel / walked in stream order
        for (size_t i = 0; i < ns; ++i)
            if (rec[i].count) (rec[i].count <= 3 ? shift : exact).push_back((uint32_t)i);
        if (!shift.empty() || !exact.empty()) {
            SDA_TRY(d_list.reserve((shift.size() + exact.size()) * 4));
            uint32_t* dl = d_list.as<uint32_t>();
            if (!shift.empty()) {
                HIP_TRY(hipMemcpyAsync(dl, shift.data(), shift.size() * 4, hipMemcpyHostToDevice, ctx.stream));
                HIP_TRY(launch_chacha_mask_shift(d_seeds.as<uint32_t>(), dl, shift.size(), d_flags.as<RejectRecord>(), dimension, mod,
                                                 zone, acc.lo.as<uint64_t>(), acc.hi.as<int64_t>(), ctx.stream));
            }
            if (!exact.empty()) {
                HIP_TRY(hipMemcpyAsync(dl + shift.size(), exact.data(), exact.size() * 4, hipMemcpyHostToDevice, ctx.stream));
                HIP_TRY(launch_chacha_mask_slow(d_seeds.as<uint32_t>(), dl + shift.size(), exact.size(), dimension, mod, zone,
                                                acc.lo.as<uint64_t>(), acc.hi.as<int64_t>(), true, ctx.stream));
            }
            HIP_TRY(hipStreamSynchronize(ctx.stream));
        }
    }
    return SDA_OK;
}

// host vectors that hold key or seed material: zeroed before the memory is released, on every exit path
template <typename T>
struct WipedVec : std::vector<T> {
    using std::vector<T>::vector;
    ~WipedVec() {
        if (!this->empty()) explicit_bzero(this->data(), this->size() * sizeof(T));
    }
};

// seed words (i64, used `as u32`, chacha.rs:62-64) -> 8 ChaCha key words (rand 0.3 from_seed: at most 8)
void seed_to_key(const int64_t* words, size_t n_words, uint32_t* key8) {
    for (int i = 0; i < 8; ++i) key8[i] = (size_t)i < n_words ? (uint32_t)(uint64_t)words[i] : 0u;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// derived scheme properties - protocol/src/crypto.rs:117-155, :67-74
// -------------------------------------------------------------------------------------------------
extern "C" uint64_t sda_scheme_input_size(const sda_sharing_scheme_t* s) {
    return !s ? 0 : sharing_is_additive(s) ? 1 : s->secret_count;
}
extern "C" uint64_t sda_scheme_output_size(const sda_sharing_scheme_t* s) { return s ? s->share_count : 0; }
extern "C" uint64_t sda_scheme_privacy_threshold(const sda_sharing_scheme_t* s) {
    return !s ? 0 : sharing_is_additive(s) ? s->share_count - 1 : s->privacy_threshold;
}
extern "C" uint64_t sda_scheme_reconstruction_threshold(const sda_sharing_scheme_t* s) {
    return !s ? 0 : sharing_is_additive(s) ? s->share_count : s->privacy_threshold + s->secret_count;
}
extern "C" int sda_masking_has_mask(const sda_masking_scheme_t* s) { return s && s->kind != SDA_MASKING_NONE; }

// -------------------------------------------------------------------------------------------------
// device memory helpers
// -------------------------------------------------------------------------------------------------
extern "C" int sda_dev_malloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return fail(SDA_ERR_INVALID_ARGUMENT, "d_ptr is NULL");
    if (sda_device_count() == 0) return fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    HIP_TRY(hipSetDevice(g_device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(SDA_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SDA_OK;
}
extern "C" int sda_dev_free(void* d_ptr) {
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return SDA_OK;
}
extern "C" int sda_dev_upload(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes) HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return SDA_OK;
}
extern "C" int sda_dev_download(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes) HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SDA_OK;
}
extern "C" int sda_dev_memset(void* d_dst, int value, size_t bytes) {
    if (bytes) HIP_TRY(hipMemset(d_dst, value, bytes));
    return SDA_OK;
}
extern "C" int sda_dev_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return SDA_OK;
}

// =================================================================================================
// ShareGenerator
// =================================================================================================
struct sda_share_generator {
    sda_sharing_scheme_t scheme;
    bool additive = true;
    uint32_t n = 0, k = 1, t = 0;        // shares, secrets per batch, random draws per batch
    ModParams mod;
    MontParams mont{0, 0};
    std::vector<uint64_t> Mmont;         // n x (k+t), Montgomery form: tss's share map (draws = values at omega_secrets^(k+1..k+t))
    std::vector<uint64_t> Msys;          // (n-t) x (k+t), Montgomery form: the systematic share map (draws = shares 0..t-1)
    bool sys_default = false;            // the device CSPRNG's draws are shares 0..t-1 (every matrix-form kernel; not the transform)
    bool sys = false;                    // ... and the handle has not been switched back (sda_share_generator_set_csprng_share_map)
    MatArg* matarg = nullptr;            // fast path (kernarg copy)
    MatArg* matarg_sys = nullptr;        // the same for Msys
    // narrow modulus (p < 2^31, the reference's own valid domain for packed Shamir): one 32-bit limb per residue
    bool n31 = false;
    N31Params n31p{};
    MatArg* matarg_n31 = nullptr;        // int32 constants, tss map
    MatArg* matarg_n31_sys = nullptr;    // int32 constants, systematic map
    PathChoice path{};                   // what select_path() decided for this scheme (path_select.hpp) ...
    PathKnobs knobs{};                   // ... under these knobs, snapshotted when the handle was created
    long knob_fft_g = 0, knob_no_lazy = 0, knob_no_side_stream = 0, knob_side_wgs = 0, knob_side_prio_high = 0, knob_ngemm_clerk_wg = 0;
    bool fast = false;
    bool l31 = false;                    // balanced-31-bit-limb kernel, matrix in the kernarg segment
    bool l31g = false;                   // the same with run-time (k, t) and the matrix in global memory (d_M)
    bool fft = false;                    // transform form (tss's own algorithm) for large tss-valid shapes
    bool mfma = false;                   // limb GEMM on the matrix cores (d_M holds the byte-reversed balanced constants)
    FftPlan fplan{};
    DevBuf d_fft;
    bool ngemm = false;                  // narrow prime (p < 2^23): the transform shapes run as a limb GEMM on the matrix cores instead
    NGemmPlan gplan{}, gplan_sys{};      // tss's share map (n rows) / the systematic one (n - t rows, shares 0..t-1 = the draws)
    DevBuf d_ngemm, d_ngemm_sys;
    DevBuf d_cw_progress;                // limb GEMM, dual-role launch: (item, row) each clerk-wave slot stopped at (ngemm_kernels.hip)
    L31Params lp{};
    Drbg drbg;
    Ctx ctx;
    DevBuf d_M, d_Msys, d_secrets, d_rand, d_out;
    // generate_combine_dev for shapes without a dual-role kernel: the clerk sum of the previous tile runs on this side stream
    // beside the share generation (fork / join with events), created on first use
    bool rust_signed = false;            // SDA_VALUES_RUST_SIGNED (additive only): additive.rs:42-47 with Rust's own `%`
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int side_stream() {
        if (aux && ev_fork && ev_join) return SDA_OK;
        // a stream of another priority class gets a hardware queue of its own (plain streams are dealt round-robin over a
        // few queues and this one landed on the default stream's: both kernels then ran back to back, rocprofv3 queue ids)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // LOW priority: the transform kernel's workgroups (8 waves + 77 KB of LDS each) must keep being dispatched first; the
        // clerk sum takes the wave slots and registers they leave (at high priority it starved them: no gain, measured)
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, knob_side_prio_high ? greatest : least);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&e0, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
        if (e != hipSuccess) {                                   // all three or none: a half-built set would fail every later call
            if (e1) (void)hipEventDestroy(e1);
            if (e0) (void)hipEventDestroy(e0);
            if (st) (void)hipStreamDestroy(st);
            return fail(SDA_ERR_HIP, "side stream setup failed: %s", hipGetErrorString(e));
        }
        aux = st; ev_fork = e0; ev_join = e1;
        return SDA_OK;
    }
};

// Builds the n x (k+t) share matrix of tss::packed::PackedSecretSharing::share:
// the polynomial f with f(w2^0) = 0, f(w2^i) = secret_i (i = 1..k), f(w2^(k+j)) = draw_j (j = 1..t),
// shares f(w3^j), j = 1..n.
static int build_packed_share_matrix(const sda_sharing_scheme_t& s, uint64_t p, std::vector<uint64_t>& out) {
    const uint64_t m = s.secret_count + s.privacy_threshold;
    const uint64_t w2 = h_canon(s.omega_secrets, p), w3 = h_canon(s.omega_shares, p);
    std::vector<uint64_t> nodes(m + 1), evals(s.share_count);
    for (uint64_t e = 0; e <= m; ++e) nodes[e] = h_powmod(w2, e, p);
    for (uint64_t j = 0; j < s.share_count; ++j) evals[j] = h_powmod(w3, j + 1, p);
    if (!h_all_distinct(nodes))
        return fail(SDA_ERR_INVALID_ARGUMENT, "omega_secrets has order <= secret_count + privacy_threshold: interpolation nodes collide");
    if (!h_lagrange_matrix_mont(nodes, evals, 0, p, out)) return fail(SDA_ERR_INVALID_ARGUMENT, "share matrix is singular");
    return SDA_OK;
}

// The systematic share map of the device CSPRNG (include/sda_hip.h, "CSPRNG share map"): the same polynomial family - degree
// <= t + k, f(1) = 0, f(w2^i) = secret_i - parametrised by its values at the first t SHARE points instead of tss's t extra
// nodes w2^(k+1..k+t): shares 0..t-1 are the draws themselves, and this (n - t) x (k + t) matrix gives the other shares from
// [secrets ; draws].  For fixed secrets both parametrisations are bijections onto the same t-dimensional family (t + k + 1
// distinct points fix a polynomial of degree <= t + k), so uniform draws give the same joint distribution of shares as tss's
// construction (packed_shamir.rs:42).  Unavailable (returns false, no error) when a share point collides with a node.
static bool build_systematic_share_matrix(const sda_sharing_scheme_t& s, uint64_t p, std::vector<uint64_t>& out) {
    const uint64_t k = s.secret_count, t = s.privacy_threshold, n = s.share_count;
    if (t == 0 || n < t) return false;
    const uint64_t w2 = h_canon(s.omega_secrets, p), w3 = h_canon(s.omega_shares, p);
    std::vector<uint64_t> nodes(k + t + 1), evals(n - t);
    for (uint64_t e = 0; e <= k; ++e) nodes[e] = h_powmod(w2, e, p);
    for (uint64_t j = 0; j < t; ++j) nodes[k + 1 + j] = h_powmod(w3, j + 1, p);
    for (uint64_t j = t; j < n; ++j) evals[j - t] = h_powmod(w3, j + 1, p);
    if (!h_all_distinct(nodes)) return false;
    return h_lagrange_matrix_mont(nodes, evals, 0, p, out);
}

// The scheme descriptor travels over the network in the reference protocol (Aggregation resource): every field is
// bounded on its own BEFORE any sum is formed, so no u64 wrap-around can slip through.
static int validate_packed(const sda_sharing_scheme_t& s) {
    if (s.secret_count < 1 || s.secret_count > 4096) return fail(SDA_ERR_INVALID_ARGUMENT, "secret_count must be in 1..4096");
    if (s.privacy_threshold > 4096) return fail(SDA_ERR_INVALID_ARGUMENT, "privacy_threshold must be <= 4096");
    if (s.share_count < 1 || s.share_count > 65535) return fail(SDA_ERR_INVALID_ARGUMENT, "share_count must be in 1..65535");
    if (s.secret_count + s.privacy_threshold > 4096)
        return fail(SDA_ERR_UNSUPPORTED, "packed scheme too large (secret_count + privacy_threshold > 4096)");
    // tss zero-extends the t + k + 1 coefficients to n + 1 evaluation points: fewer shares than t + k can never be
    // reconstructed (reconstruct_limit() = t + k)
    if (s.share_count < s.secret_count + s.privacy_threshold)
        return fail(SDA_ERR_INVALID_ARGUMENT, "share_count must be >= secret_count + privacy_threshold (the reconstruction threshold)");
    // privacy_threshold == 0 is accepted (tss does when secret_count + 1 is a power of two): a sharing without privacy
    if (s.modulus < 3 || (s.modulus & 1) == 0 || !h_is_prime((uint64_t)s.modulus))
        return fail(SDA_ERR_INVALID_ARGUMENT, "prime_modulus %lld is not an odd prime", (long long)s.modulus);
    return SDA_OK;
}

// Constants of the balanced-limb kernel: matrix entries in Montgomery form with R = 2^62, centred to
// (-p/2, p/2] and split into balanced limbs m1 * 2^31 + m0, m0 in [-2^30, 2^30).
static void l31_pack_matrix(const std::vector<uint64_t>& Mm, uint64_t p, unsigned r_bits, std::vector<uint64_t>& packed) {
    const uint64_t B = 1ull << 31;
    packed.assign(Mm.size() + 7, 0);                                    // zero entries past the end: the run-time kernels read up to six (l31_dot_rt)
    // Mm holds M * 2^64: bring it to M * 2^r_bits (62: divide by 4; 93: multiply by 2^29)
    const uint64_t adjust = r_bits == 62 ? h_powmod(4 % p, p - 2, p) : h_powmod(2, r_bits - 64, p);
    for (size_t i = 0; i < Mm.size(); ++i) {
        const uint64_t mr = h_mulmod(Mm[i], adjust, p);                 // M * 2^r_bits mod p
        __int128 c = mr > (p - 1) / 2 ? (__int128)mr - (__int128)p : (__int128)mr;
        int64_t c64 = (int64_t)c;
        int64_t m0 = (int64_t)(((uint64_t)c64 & (B - 1)));
        if (m0 >= (int64_t)(B >> 1)) m0 -= (int64_t)B;
        const int64_t m1 = (c64 - m0) / (int64_t)B;
        packed[i] = (uint64_t)(uint32_t)(int32_t)m0 | ((uint64_t)(uint32_t)(int32_t)m1 << 32);
    }
}

static int l31_params(uint64_t p, L31Params& lp);

// The three-digit kernels put a remainder of one term into an 8-term last group (k + t = 15: 7 + 8).  Eight products of two
// limbs can pass a signed 64-bit column only when all eight constant limbs are -2^30; checked here on the actual constants
// for ANY values (|limb| <= 2^30) plus what the normalisation carries in: false -> the handle takes another kernel.
static bool l31_eight_term_group_ok(const std::vector<uint64_t>& Mm, uint32_t kt, uint64_t p) {
    if (kt <= 7 || kt % 7 != 1 || Mm.empty()) return true;
    std::vector<uint64_t> packed;
    l31_pack_matrix(Mm, p, 93, packed);
    for (size_t r = 0; r + kt <= Mm.size(); r += kt) {
        uint64_t s0 = 0, s1 = 0;
        for (uint32_t i = kt - 8; i < kt; ++i) {
            const int64_t m0 = (int32_t)(uint32_t)packed[r + i], m1 = (int32_t)(uint32_t)(packed[r + i] >> 32);
            s0 += (uint64_t)(m0 < 0 ? -m0 : m0);
            s1 += (uint64_t)(m1 < 0 ? -m1 : m1);
        }
        if ((s0 << 30) + (1ull << 32) >= (1ull << 63) || (s1 << 30) + (1ull << 33) >= (1ull << 63)) return false;
    }
    return true;
}

// The three-digit kernels of 9 .. 12 terms can run a whole dot product as ONE group when no column can leave a signed 64-bit
// register: |column| <= 2^30 x sum |constant limb| for ANY values (limbs of magnitude <= 2^30), so sum |limb| x 2^30 (+ slack)
// must stay below 2^63 for the m0 and the m1 limbs of every row.  Checked on the actual constants (R = 2^93).
static bool l31_wide_group_ok(const std::vector<uint64_t>& Mm, uint32_t kt, uint64_t p) {
    if (kt < 9 || kt > 12 || Mm.empty()) return false;
    std::vector<uint64_t> packed;
    l31_pack_matrix(Mm, p, 93, packed);
    for (size_t r = 0; r + kt <= Mm.size(); r += kt) {
        uint64_t s0 = 0, s1 = 0;
        for (uint32_t i = 0; i < kt; ++i) {
            const int64_t m0 = (int32_t)(uint32_t)packed[r + i], m1 = (int32_t)(uint32_t)(packed[r + i] >> 32);
            s0 += (uint64_t)(m0 < 0 ? -m0 : m0);
            s1 += (uint64_t)(m1 < 0 ? -m1 : m1);
        }
        if ((s0 << 30) + (1ull << 33) >= (1ull << 63) || (s1 << 30) + (1ull << 33) >= (1ull << 63)) return false;
    }
    return true;
}

// The Karatsuba form of the wide group (l31_dot3_wide_k): each HALF of the terms keeps one cross column M - C0 - C2, which must fit a
// signed 64-bit register for any values (limbs of magnitude <= 2^30): sum over the half of (|m0| + |m1|) x 2^30 + slack < 2^63.
static bool l31_karatsuba_ok(const std::vector<uint64_t>& Mm, uint32_t kt, uint64_t p) {
    if (!l31_wide_group_ok(Mm, kt, p)) return false;
    std::vector<uint64_t> packed;
    l31_pack_matrix(Mm, p, 93, packed);
    const uint32_t half = (kt + 1) / 2;
    for (size_t r = 0; r + kt <= Mm.size(); r += kt)
        for (uint32_t lo = 0; lo < kt; lo += half) {
            uint64_t s = 0;
            for (uint32_t i = lo; i < lo + half && i < kt; ++i) {
                const int64_t m0 = (int32_t)(uint32_t)packed[r + i], m1 = (int32_t)(uint32_t)(packed[r + i] >> 32);
                s += (uint64_t)(m0 < 0 ? -m0 : m0) + (uint64_t)(m1 < 0 ? -m1 : m1);
            }
            if ((s << 30) + (1ull << 33) >= (1ull << 63)) return false;
        }
    return true;
}

// one matrix in the form the limb-31 kernels take: the kernarg copy (compiled / run-time (k, t) kernels) or device memory
static int l31_place_matrix(sda_share_generator* g, const std::vector<uint64_t>& Mm, MatArg*& arg, DevBuf& dev) {
    std::vector<uint64_t> packed;
    // the radix follows the shape: compiled three-digit instances and every run-time shape of more than eight terms carry R = 2^93
    l31_pack_matrix(Mm, g->mod.m, g->l31g ? packed_l31_rt_r_bits(g->k + g->t) : packed_l31_r_bits(g->k, g->t), packed);
    if (g->l31g) {                                                      // matrix in global memory
        SDA_TRY(dev.reserve(packed.size() * 8));
        HIP_TRY(hipMemcpy(dev.p, packed.data(), packed.size() * 8, hipMemcpyHostToDevice));
        return SDA_OK;
    }
    arg = new (std::nothrow) MatArg();
    if (!arg) return fail(SDA_ERR_ALLOC, "out of memory");
    memset(arg, 0, sizeof(MatArg));
    memcpy(arg->e, packed.data(), Mm.size() * 8);
    return SDA_OK;
}

// Constants of the narrow kernels: centred representatives of M * 2^32 mod p as int32, rows back to back in a MatArg
static int n31_place_matrix(const std::vector<uint64_t>& Mm, uint64_t p, MatArg*& arg) {
    arg = new (std::nothrow) MatArg();
    if (!arg) return fail(SDA_ERR_ALLOC, "out of memory");
    memset(arg, 0, sizeof(MatArg));
    int32_t* e = reinterpret_cast<int32_t*>(arg->e);
    uint64_t inv32;                                                     // 2^-32 mod p: Mm holds M * 2^64
    if (!h_invmod((1ull << 32) % p, p, inv32)) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus not odd");
    for (size_t i = 0; i < Mm.size(); ++i) {
        const uint64_t mr = h_mulmod(Mm[i], inv32, p);                  // M * 2^32 mod p
        e[i] = mr > (p - 1) / 2 ? (int32_t)((int64_t)mr - (int64_t)p) : (int32_t)mr;
    }
    return SDA_OK;
}
static int build_n31(sda_share_generator* g) {
    const uint64_t p = g->mod.m;
    uint64_t inv;
    if (!h_invmod(p, 1ull << 32, inv)) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus not invertible mod 2^32");
    g->n31p.p = (uint32_t)p; g->n31p.pinv = (uint32_t)((1ull << 32) - inv); g->n31p.h = (uint32_t)((p + 1) / 2); g->n31p.pad = 0;
    SDA_TRY(n31_place_matrix(g->Mmont, p, g->matarg_n31));
    if (g->sys_default) SDA_TRY(n31_place_matrix(g->Msys, p, g->matarg_n31_sys));
    g->n31 = true;
    return SDA_OK;
}

static int build_l31(sda_share_generator* g) {
    SDA_TRY(l31_params(g->mod.m, g->lp));
    if (g->l31 && packed_l31_three_digit_compiled(g->k, g->t))          // both maps, or the 7 + rest form serves the handle
        g->lp.wide = !knob(KNOB_NO_WIDE_GROUP) && l31_wide_group_ok(g->Mmont, g->k + g->t, g->mod.m) && (!g->sys_default || l31_wide_group_ok(g->Msys, g->k + g->t, g->mod.m)) ? 1u : 0u;
        if (g->lp.wide && !knob(KNOB_NO_KARATSUBA) && l31_karatsuba_ok(g->Mmont, g->k + g->t, g->mod.m) && (!g->sys_default || l31_karatsuba_ok(g->Msys, g->k + g->t, g->mod.m)))
            g->lp.wide = 2u;                                            // three multiply-adds per term (round 6)
    SDA_TRY(l31_place_matrix(g, g->Mmont, g->matarg, g->d_M));
    if (g->sys_default) SDA_TRY(l31_place_matrix(g, g->Msys, g->matarg_sys, g->d_Msys));
    return SDA_OK;
}

// l31 constants without the matrix (shared by the limb-31 and the transform paths)
static int l31_params(uint64_t p, L31Params& lp) {
    const uint64_t B = 1ull << 31;
    uint64_t inv;
    if (!h_invmod(p % B, B, inv)) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus not invertible mod 2^31");
    lp.p = p; lp.p2 = 2 * p; lp.h = (p + 1) / 2;
    lp.p0 = (int32_t)(p % B); lp.p1 = (int32_t)(p >> 31);
    lp.pinvB = (uint32_t)((B - inv) % B); lp.wide = 0;
    lp.np = (uint64_t)0 - p; lp.np2 = (uint64_t)0 - 2 * p;
    return SDA_OK;
}

// tss's transform structure applies when k + t + 1 = 2^a = ord(omega_secrets) and n + 1 = 3^b = ord(omega_shares)
// (SURVEY.md App. B); the kernel also needs the group's values in LDS (p < 2^62 holds for every modulus the library takes).
// Two workgroups per CU (80 KB each) when a group of 8 batches fits, with the twiddle tables in LDS too if there is room.
static bool fft_narrow(uint64_t p, const PathKnobs& kn) { return p < (1ull << 30) && !kn.no_narrow; }   // 4p < 2^32: the kernel's lazy ranges in 32 bits
static bool fft_shape(const sda_share_generator* g, const PathKnobs& kn, uint32_t& a, uint32_t& b, uint32_t& G, uint32_t& tw_lds) {
    const bool narrow = fft_narrow(g->mod.m, kn);
    const uint64_t p = g->mod.m, m2 = (uint64_t)g->k + g->t + 1, m3 = (uint64_t)g->n + 1;
    if (p >= (1ull << 62)) return false;
    a = 0; while ((1ull << a) < m2) ++a;
    if ((1ull << a) != m2) return false;
    uint64_t q = 1; b = 0; while (q < m3) { q *= 3; ++b; }
    if (q != m3 || b < 2 || m3 > 19683 || m2 > 4096 || m2 > m3) return false;
    const uint64_t w2 = h_canon(g->scheme.omega_secrets, p), w3 = h_canon(g->scheme.omega_shares, p);
    if (h_powmod(w2, m2, p) != 1 || h_powmod(w2, m2 / 2, p) == 1) return false;      // order exactly 2^a
    if (h_powmod(w3, m3, p) != 1 || h_powmod(w3, m3 / 3, p) == 1) return false;      // order exactly 3^b
    const size_t half_cu = 80 * 1024, whole_cu = 160 * 1024;
    // (16 batches in one 1024-thread workgroup per CU - whole 128-byte lines per clerk row - was measured: 41.7 ms against 36.0
    // per 500-participant tile of PSS_155_728_100, its barriers no longer hidden by a second workgroup, and the same
    // WRITE_SIZE: the 1.35x write amplification of round 2 came from write-back stores, not from half lines - the
    // non-temporal stores of the last pass bring it to 1.02x in either form.)
    // the most batches per workgroup (8, 4, 2, 1: a power of two inside one CSPRNG block group of 8) that still leaves two
    // workgroups per CU, twiddles in LDS when they fit too; a single batch may take the whole CU (PSS_155_19682_100: 157 KB)
    G = 0;
    // narrow values are half the size: 16 batches per workgroup (whole 128-byte lines per clerk row) still leave two workgroups
    // per CU - PSS_155_728_100 over 746497: 25.5 Gelem/s against 21.0 with 8 (interleaved on one box, profiles/r04/narrow_bench.txt)
    for (uint32_t cand = narrow ? 16 : 8; cand >= 1 && !G; cand >>= 1)
        for (uint32_t tw = 2; tw-- > 0 && !G;)
            if (fft_lds_bytes((uint32_t)m2, (uint32_t)m3, cand, tw != 0, narrow) <= half_cu) { G = cand; tw_lds = tw; }
    if (!G && fft_lds_bytes((uint32_t)m2, (uint32_t)m3, 1, false, narrow) <= whole_cu) { G = 1; tw_lds = 0; }
    if (const long fg = g->knob_fft_g) {                               // A/B only: fewer batches per workgroup
        const uint32_t want = (uint32_t)fg;
        if (want == 1 || want == 2 || want == 4 || want == 8 || want == 16) {
            if (fft_lds_bytes((uint32_t)m2, (uint32_t)m3, want, true, narrow) <= half_cu) { G = want; tw_lds = 1; }
            else if (fft_lds_bytes((uint32_t)m2, (uint32_t)m3, want, false, narrow) <= whole_cu) { G = want; tw_lds = 0; }
        }
    }
    return G != 0;
}

// a constant and its Shoup companion floor(w 2^64 / p)
static void shoup_pair(uint64_t w, uint64_t p, uint64_t& out_w, uint64_t& out_s) {
    out_w = w;
    out_s = (uint64_t)((((u128)w) << 64) / p);
}

// the limb-GEMM kernel's constants: centred Montgomery-form (R = 2^64) entries in balanced base-256 digits (the kernel cuts its
// Toeplitz rows out of them), [n][8 * ceil((k + t) / 8)] zero padded
static int mfma_place_matrix(sda_share_generator* g, const std::vector<uint64_t>& Mm, DevBuf& dev) {
    const uint32_t kt = g->k + g->t, width = 8 * ((kt + 7) / 8);
    const size_t rows = Mm.size() / kt;
    std::vector<uint64_t> tab(rows * width, 0);
    for (size_t j = 0; j < rows; ++j)
        for (uint32_t i = 0; i < kt; ++i) {
            uint64_t m = Mm[j * kt + i];
            if (m > (g->mod.m >> 1)) m -= g->mod.m;                          // centred representative, two's complement
            const uint64_t bal = (m + 0x8080808080808080ull) ^ 0x8080808080808080ull;
            tab[j * width + i] = bal;
        }
    SDA_TRY(dev.reserve(tab.size() * 8 + 8));
    HIP_TRY(hipMemcpy(dev.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    return SDA_OK;
}
static int build_mfma(sda_share_generator* g) {
    SDA_TRY(mfma_place_matrix(g, g->Mmont, g->d_M));
    if (g->sys_default) SDA_TRY(mfma_place_matrix(g, g->Msys, g->d_Msys));
    return SDA_OK;
}

static int build_fft(sda_share_generator* g, uint32_t a, uint32_t b, uint32_t G, uint32_t tw_lds) {
    const uint64_t p = g->mod.m, m2 = (uint64_t)g->k + g->t + 1, m3 = (uint64_t)g->n + 1;
    const uint64_t w2 = h_canon(g->scheme.omega_secrets, p), w3 = h_canon(g->scheme.omega_shares, p);
    uint64_t w2i, m2i;
    if (!h_invmod(w2, p, w2i) || !h_invmod(m2 % p, p, m2i)) return fail(SDA_ERR_INVALID_ARGUMENT, "omega_secrets is not invertible");
    const bool narrow = fft_narrow(p, g->knobs);
    FftPlan& F = g->fplan;
    F.narrow = narrow ? 1u : 0u;
    if (narrow) {                                               // the same tables as (uint32 w, uint32 floor(w 2^32 / p)) pairs
        std::vector<uint32_t> tab(2 * (m3 + m2 / 2));
        uint64_t x = 1;
        for (uint64_t j = 0; j < m3; ++j) { tab[2 * j] = (uint32_t)x; tab[2 * j + 1] = (uint32_t)((x << 32) / p); x = h_mulmod(x, w3, p); }
        x = 1;
        for (uint64_t j = 0; j < m2 / 2; ++j) { tab[2 * (m3 + j)] = (uint32_t)x; tab[2 * (m3 + j) + 1] = (uint32_t)((x << 32) / p); x = h_mulmod(x, w2i, p); }
        SDA_TRY(g->d_fft.reserve(tab.size() * 4));
        HIP_TRY(hipMemcpy(g->d_fft.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint64_t> tab(2 * (m3 + m2 / 2));           // [radix-3 twiddles | radix-2 twiddles], (w, companion) pairs
        uint64_t x = 1;
        for (uint64_t j = 0; j < m3; ++j) { shoup_pair(x, p, tab[2 * j], tab[2 * j + 1]); x = h_mulmod(x, w3, p); }
        x = 1;
        for (uint64_t j = 0; j < m2 / 2; ++j) { shoup_pair(x, p, tab[2 * (m3 + j)], tab[2 * (m3 + j) + 1]); x = h_mulmod(x, w2i, p); }
        SDA_TRY(g->d_fft.reserve(tab.size() * 8));
        HIP_TRY(hipMemcpy(g->d_fft.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    }
    F.k = g->k; F.t = g->t; F.n = g->n; F.m2 = (uint32_t)m2; F.a = a; F.m3 = (uint32_t)m3; F.b = b; F.G = G;
    F.lgG = G == 16 ? 4 : G == 8 ? 3 : G == 4 ? 2 : G == 2 ? 1 : 0;      // 16: narrow kernel only (A/B knob SDA_FFT_G)
    F.tw_lds = tw_lds;
    F.nz_mask = 0;
    for (uint32_t e0 = 0; e0 < 3; ++e0)
        for (uint32_t e1 = 0; e1 < 3; ++e1)
            if (e1 * (m3 / 9) + e0 * (m3 / 3) < m2) F.nz_mask |= 1u << (3 * e0 + e1);
    F.tw3 = g->d_fft.as<uint64_t>();
    F.tw2 = narrow ? reinterpret_cast<const uint64_t*>(g->d_fft.as<uint32_t>() + 2 * m3) : g->d_fft.as<uint64_t>() + 2 * m3;
    shoup_pair(h_powmod(w3, m3 / 3, p), p, F.omega, F.omega_s);
    shoup_pair(m2i, p, F.scale, F.scale_s);
    if (narrow) { F.omega_s = (F.omega << 32) / p; F.scale_s = (F.scale << 32) / p; }
    F.one_s = narrow ? (1ull << 32) / p : 0;
    F.lazy = narrow && (4ull * b + 4) * p < (1ull << 32) && !g->knob_no_lazy ? 1u : 0u;
    F.magic_k1 = (uint32_t)(0x100000000ull / ((uint64_t)g->k + 1)) + 1u;                 // k + 1 >= 2
    F.magic_t = g->t > 1 ? (uint32_t)(0x100000000ull / g->t) + 1u : 0u;
    F.want_threads = (uint32_t)knob(KNOB_FFT_THREADS);
    F.no_xcd_map = knob(KNOB_NO_XCD_MAP) ? 1u : 0u;
    return SDA_OK;
}

// The narrow limb GEMM's constants (ngemm_kernels.hip): the plain matrix M = Mmont 2^-64, centred, in three balanced base-256
// digits, laid out as the A fragments of v_mfma_i32_16x16x64_i8 - [row tile][64-term step][digit][lane = row | g << 4][16 terms] -
// zero beyond n rows / k + t terms; and c_j = 256^j 2^32 mod p (centred) for the epilogue's Montgomery operand.
static int build_ngemm_plan(sda_share_generator* g, const std::vector<uint64_t>& Mm, uint32_t rows, DevBuf& dev, NGemmPlan& P) {
    const uint64_t p = g->mod.m;
    const uint32_t kt = g->k + g->t, ks = packed_ngemm_steps(g->k, g->t), tiles = (rows + 15) / 16;
    uint64_t inv64, inv;
    if (!h_invmod(h_powmod(2, 64, p), p, inv64) || !h_invmod(p, 1ull << 32, inv)) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus not odd");
    const size_t tile_bytes = ngemm_tile_bytes(ks);
    std::vector<uint8_t> A((size_t)tiles * tile_bytes, 0);
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t i = 0; i < kt; ++i) {
            const uint64_t m = h_mulmod(Mm[(size_t)r * kt + i], inv64, p);
            const int64_t x = m > (p - 1) / 2 ? (int64_t)m - (int64_t)p : (int64_t)m;
            const uint32_t d = ((uint32_t)(int32_t)x + 0x00808080u) ^ 0x00808080u;
            const uint32_t rt = r >> 4, row = r & 15u, step = i >> 6, gq = (i >> 4) & 3u, j = i & 15u;
            for (uint32_t la = 0; la < 3; ++la)
                A[(size_t)rt * tile_bytes + (((size_t)step * 3 + la) * 64 + (row | gq << 4)) * 16 + j] = (uint8_t)(d >> (8 * la));
        }
    SDA_TRY(dev.reserve(A.size() + 16));
    if (!A.empty()) HIP_TRY(hipMemcpy(dev.p, A.data(), A.size(), hipMemcpyHostToDevice));
    P.k = g->k; P.t = g->t; P.n = rows; P.ks = ks; P.row_tiles = tiles;
    P.np.p = (uint32_t)p; P.np.pinv = (uint32_t)inv /* +p^-1 mod 2^32: ng_redc subtracts */; P.np.h = (uint32_t)((p + 1) / 2); P.np.pad = 0;
    uint64_t c = (1ull << 32) % p;
    for (int j = 0; j < 5; ++j) {
        P.c[j] = c > (p - 1) / 2 ? (int32_t)((int64_t)c - (int64_t)p) : (int32_t)c;
        c = h_mulmod(c, 256, p);
    }
    P.A = dev.as<uint8_t>();
    return SDA_OK;
}
static int build_ngemm(sda_share_generator* g) {
    SDA_TRY(build_ngemm_plan(g, g->Mmont, g->n, g->d_ngemm, g->gplan));
    // with the library's own randomness the draws ARE shares 0..t-1 (the systematic share map of the matrix-form kernels):
    // t of the n rows cost nothing (PSS_155_728_100: 155 of 728)
    if (g->Msys.empty()) g->sys_default = build_systematic_share_matrix(g->scheme, g->mod.m, g->Msys);
    if (g->sys_default) SDA_TRY(build_ngemm_plan(g, g->Msys, g->n - g->t, g->d_ngemm_sys, g->gplan_sys));
    g->ngemm = true;
    return SDA_OK;
}

// the selection knobs as they stand now (include/sda_hip_debug.h): read ONCE per handle, so that a knob changed later cannot
// switch a live handle to another kernel family or share map
static PathKnobs snapshot_knobs() {
    PathKnobs kn;
    kn.force_generic = knob(KNOB_FORCE_GENERIC) != 0; kn.force_mont64 = knob(KNOB_FORCE_MONT64) != 0;
    kn.force_fft = knob(KNOB_FORCE_FFT) != 0; kn.force_mfma = knob(KNOB_FORCE_MFMA) != 0; kn.no_mfma = knob(KNOB_NO_MFMA) != 0;
    kn.no_narrow = knob(KNOB_NO_NARROW) != 0; kn.no_ngemm = knob(KNOB_NO_NGEMM) != 0;
    return kn;
}
static void snapshot_call_knobs(sda_share_generator* g) {
    g->knob_fft_g = knob(KNOB_FFT_G); g->knob_no_lazy = knob(KNOB_NO_LAZY); g->knob_no_side_stream = knob(KNOB_NO_SIDE_STREAM);
    g->knob_side_wgs = knob(KNOB_SIDE_STREAM_WGS); g->knob_side_prio_high = knob(KNOB_SIDE_STREAM_PRIORITY_HIGH);
    g->knob_ngemm_clerk_wg = knob(KNOB_NGEMM_CLERK_WG);
}
// what select_path() needs to know about the constants: host arithmetic only (g->Mmont must be built)
static PathFacts path_facts(const sda_share_generator* g, const PathKnobs& kn) {
    PathFacts f;
    uint32_t a = 0, b = 0, G = 0, tw = 0;
    f.transform_shape = fft_shape(g, kn, a, b, G, tw);
    if (packed_l31_path_available(g->k, g->t, g->n) && packed_l31_three_digit_compiled(g->k, g->t)) {
        // the 8-term last group of a three-digit shape is admitted on the constants of BOTH share maps
        std::vector<uint64_t> sys_tmp;
        f.eight_term_ok = l31_eight_term_group_ok(g->Mmont, g->k + g->t, g->mod.m);
        if (f.eight_term_ok && build_systematic_share_matrix(g->scheme, g->mod.m, sys_tmp))
            f.eight_term_ok = l31_eight_term_group_ok(sys_tmp, g->k + g->t, g->mod.m);
    }
    return f;
}

extern "C" int sda_share_generator_new(const sda_sharing_scheme_t* scheme, sda_share_generator_t** out) {
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    SDA_TRY(check_scheme_kind(scheme));
    sda_share_generator* g = new (std::nothrow) sda_share_generator();
    if (!g) return fail(SDA_ERR_ALLOC, "out of memory");
    g->scheme = *scheme;
    int st = make_mod(scheme->modulus, g->mod);
    if (st == SDA_OK) st = g->drbg.init();
    if (st == SDA_OK) {
        if (sharing_is_additive(scheme)) {
            if (scheme->share_count < 1 || scheme->share_count > 65535)
                st = fail(SDA_ERR_INVALID_ARGUMENT, "share_count must be in 1..65535");
            g->additive = true;
            g->n = (uint32_t)scheme->share_count; g->k = 1; g->t = g->n - 1;
        } else {
            st = validate_packed(*scheme);
            if (st == SDA_OK) {
                g->additive = false;
                g->n = (uint32_t)scheme->share_count; g->k = (uint32_t)scheme->secret_count; g->t = (uint32_t)scheme->privacy_threshold;
                MontCtx mc = h_mont_ctx(g->mod.m);
                g->mont.p = mc.p; g->mont.pinv = mc.pinv;
                st = build_packed_share_matrix(*scheme, g->mod.m, g->Mmont);
            }
        }
    }
    if (st == SDA_OK) st = g->ctx.init();
    if (st == SDA_OK && !g->additive) {
        // ONE decision (path_select.hpp), then build what it names.  The wide family serves every call; a narrow family on top of
        // it (p < 2^31) serves the ChaCha20 / injected-randomness calls.
        g->knobs = snapshot_knobs();
        snapshot_call_knobs(g);
        g->path = select_path(g->k, g->t, g->n, g->mod.m, path_facts(g, g->knobs), g->knobs);
        g->fft = g->path.wide == WIDE_FFT; g->mfma = g->path.wide == WIDE_MFMA; g->l31 = g->path.wide == WIDE_L31;
        g->l31g = g->path.wide == WIDE_L31_GLOBAL; g->fast = g->path.wide == WIDE_MONT64;
        // the systematic share map (draws = shares 0..t-1) exists for every matrix-form family, not for the transform
        if (!g->fft || g->path.narrow == NARROW_NGEMM) g->sys_default = build_systematic_share_matrix(*scheme, g->mod.m, g->Msys);
        switch (g->path.wide) {
            case WIDE_FFT: {
                uint32_t fa = 0, fb = 0, fG = 0, ftw = 0;
                if (!fft_shape(g, g->knobs, fa, fb, fG, ftw)) st = fail(SDA_ERR_STATE, "transform plan vanished between selection and build");
                else st = build_fft(g, fa, fb, fG, ftw);
                break;
            }
            case WIDE_MFMA: st = build_mfma(g); break;
            case WIDE_L31:
            case WIDE_L31_GLOBAL: st = build_l31(g); break;
            case WIDE_MONT64:
                g->matarg = new (std::nothrow) MatArg();
                if (g->sys_default) g->matarg_sys = new (std::nothrow) MatArg();
                if (!g->matarg || (g->sys_default && !g->matarg_sys)) st = fail(SDA_ERR_ALLOC, "out of memory");
                else {
                    memset(g->matarg, 0, sizeof(MatArg));
                    memcpy(g->matarg->e, g->Mmont.data(), g->Mmont.size() * 8);
                    if (g->sys_default) {
                        memset(g->matarg_sys, 0, sizeof(MatArg));
                        memcpy(g->matarg_sys->e, g->Msys.data(), g->Msys.size() * 8);
                    }
                }
                break;
            default:
                st = g->d_M.reserve(g->Mmont.size() * 8);
                if (st == SDA_OK && hipMemcpy(g->d_M.p, g->Mmont.data(), g->Mmont.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
                    st = fail(SDA_ERR_HIP, "uploading the share matrix failed");
                if (st == SDA_OK && g->sys_default) {
                    st = g->d_Msys.reserve(g->Msys.size() * 8 + 8);
                    if (st == SDA_OK && hipMemcpy(g->d_Msys.p, g->Msys.data(), g->Msys.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
                        st = fail(SDA_ERR_HIP, "uploading the systematic share matrix failed");
                }
                break;
        }
        if (st == SDA_OK && g->path.narrow == NARROW_N31) st = build_n31(g);
        if (st == SDA_OK && g->path.narrow == NARROW_NGEMM) st = build_ngemm(g);
        g->sys = g->sys_default;
    }
    if (st != SDA_OK) { sda_share_generator_free(g); return st; }
    *out = g;
    return SDA_OK;
}

#ifdef SDA_TEST_HOOKS
// The decision table without a device (tests/test_path_select.py runs it on a machine with no GPU): validates the scheme, builds
// the host-side facts, calls select_path() under the NAMED knobs (comma separated, NULL / "" = defaults; the process-wide knob
// state is not read) and describes the choice and what each kind of call would run.
extern "C" int sda_debug_select_path(const sda_sharing_scheme_t* scheme, const char* knobs, char* out, size_t cap) {
    if (!out || cap < 8) return fail(SDA_ERR_INVALID_ARGUMENT, "out needs a buffer");
    out[0] = 0;
    SDA_TRY(check_scheme_kind(scheme));
    if (sharing_is_additive(scheme)) { snprintf(out, cap, "wide=additive narrow=none call20=additive fused20=additive"); return SDA_OK; }
    SDA_TRY(validate_packed(*scheme));
    PathKnobs kn;
    for (const char* q = knobs ? knobs : ""; *q;) {
        const char* e = strchr(q, ',');
        const size_t len = e ? (size_t)(e - q) : strlen(q);
        const std::string name(q, len);
        if (name == "SDA_FORCE_GENERIC") kn.force_generic = true;
        else if (name == "SDA_FORCE_MONT64") kn.force_mont64 = true;
        else if (name == "SDA_FORCE_FFT") kn.force_fft = true;
        else if (name == "SDA_FORCE_MFMA") kn.force_mfma = true;
        else if (name == "SDA_NO_MFMA") kn.no_mfma = true;
        else if (name == "SDA_NO_NARROW") kn.no_narrow = true;
        else if (name == "SDA_NO_NGEMM") kn.no_ngemm = true;
        else if (!name.empty()) return fail(SDA_ERR_INVALID_ARGUMENT, "%s is not a selection knob", name.c_str());
        q += len + (e ? 1 : 0);
    }
    sda_share_generator* g = new (std::nothrow) sda_share_generator();
    if (!g) return fail(SDA_ERR_ALLOC, "out of memory");
    g->scheme = *scheme; g->additive = false;
    g->n = (uint32_t)scheme->share_count; g->k = (uint32_t)scheme->secret_count; g->t = (uint32_t)scheme->privacy_threshold;
    int st = make_mod(scheme->modulus, g->mod);
    if (st == SDA_OK) st = build_packed_share_matrix(*scheme, g->mod.m, g->Mmont);
    if (st == SDA_OK) {
        const PathFacts f = path_facts(g, kn);
        const PathChoice c = select_path(g->k, g->t, g->n, g->mod.m, f, kn);
        snprintf(out, cap, "wide=%s narrow=%s r_bits=%u call20=%s call12=%s injected=%s fused20=%s fused12=%s transform_shape=%d eight_term_ok=%d",
                 wide_name(c.wide), narrow_name(c.narrow), c.wide == WIDE_L31 ? c.l31_r_bits : 0u, family_name(path_for_call(c, false, 20)),
                 family_name(path_for_call(c, false, 12)), family_name(path_for_call(c, true, 20)),
                 fused_for_call(c, 20) == FAM_GENERIC ? "none" : family_name(fused_for_call(c, 20)),
                 fused_for_call(c, 12) == FAM_GENERIC ? "none" : family_name(fused_for_call(c, 12)), (int)f.transform_shape, (int)f.eight_term_ok);
    }
    sda_share_generator_free(g);
    return st;
}
#endif

extern "C" void sda_share_generator_free(sda_share_generator_t* g) {
    if (!g) return;
    if (g->ctx.device >= 0) (void)hipSetDevice(g->ctx.device);
    g->d_M.release(); g->d_Msys.release(); g->d_fft.release(); g->d_ngemm.release(); g->d_ngemm_sys.release(); g->d_cw_progress.release(); g->d_secrets.wipe_release(); g->d_rand.wipe_release(); g->d_out.wipe_release();
    if (g->aux) { (void)hipStreamSynchronize(g->aux); (void)hipStreamDestroy(g->aux); }
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    g->ctx.destroy();
    g->drbg.wipe();
    delete g->matarg;
    delete g->matarg_sys;
    delete g->matarg_n31;
    delete g->matarg_n31_sys;
    delete g;
}

extern "C" uint64_t sda_share_generator_share_count(const sda_share_generator_t* g) { return g ? g->n : 0; }
extern "C" uint64_t sda_share_generator_batch_count(const sda_share_generator_t* g, size_t len) {
    return g ? (len + g->k - 1) / g->k : 0;                                   // batched.rs:23
}
extern "C" uint64_t sda_share_generator_rand_count(const sda_share_generator_t* g, size_t len) {
    return g ? sda_share_generator_batch_count(g, len) * g->t : 0;
}
extern "C" int sda_share_generator_set_drbg_key(sda_share_generator_t* g, const uint8_t key[32]) {
    if (!g || !key) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    g->drbg.set_stream_key(key);
    return SDA_OK;
}
extern "C" int sda_share_generator_set_drbg_master_key(sda_share_generator_t* g, const uint8_t key[32]) {
    if (!g || !key) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    g->drbg.set_master_key(key);
    return SDA_OK;
}
static int check_value_mode(int mode) {
    if (mode != SDA_VALUES_CANONICAL && mode != SDA_VALUES_RUST_SIGNED) return fail(SDA_ERR_INVALID_ARGUMENT, "unknown value mode %d", mode);
    return SDA_OK;
}
static const char* kNoSignedPacked =
    "SDA_VALUES_RUST_SIGNED covers the arithmetic visible in the reference (additive sharing, the combiner, the masks); packed "
    "Shamir's signed representatives are tss's, an un-vendored crate: compare those modulo the prime";

extern "C" int sda_share_generator_set_value_mode(sda_share_generator_t* g, int mode) {
    if (!g) return fail(SDA_ERR_INVALID_ARGUMENT, "generator is NULL");
    SDA_TRY(check_value_mode(mode));
    if (mode == SDA_VALUES_RUST_SIGNED && !g->additive) return fail(SDA_ERR_UNSUPPORTED, "%s", kNoSignedPacked);
    g->rust_signed = mode == SDA_VALUES_RUST_SIGNED;
    return SDA_OK;
}

extern "C" int sda_share_generator_set_drbg_rounds(sda_share_generator_t* g, int rounds) {
    if (!g) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    return g->drbg.set_rounds(rounds);
}

// the map the device CSPRNG's draws go through on the NEXT call: the systematic one where the handle has it, except on a
// transform-shape handle under ChaCha12 / ChaCha8 (A/B only) - the limb GEMM draws with ChaCha20 only, and the transform
// kernel that serves those calls computes tss's own map
static bool effective_sys(const sda_share_generator_t* g) {
    return g->sys && !g->additive && g->t > 0 && !(g->fft && g->drbg.rounds != 20);
}
extern "C" int sda_share_generator_csprng_share_map(const sda_share_generator_t* g) {
    return g && effective_sys(g) ? SDA_SHARE_MAP_SYSTEMATIC : SDA_SHARE_MAP_TSS_NODES;
}
// "wide[+narrow]" as select_path() chose for this scheme, e.g. "l31", "fft+ngemm", "l31+n31", "additive"
extern "C" const char* sda_share_generator_path_name(const sda_share_generator_t* g) {
    static thread_local char buf[48];
    if (!g) return "";
    if (g->additive) return "additive";
    if (g->path.narrow == NARROW_NONE) return wide_name(g->path.wide);
    snprintf(buf, sizeof buf, "%s+%s", wide_name(g->path.wide), narrow_name(g->path.narrow));
    return buf;
}
extern "C" int sda_share_generator_set_csprng_share_map(sda_share_generator_t* g, int map) {
    if (!g) return fail(SDA_ERR_INVALID_ARGUMENT, "generator is NULL");
    if (map != SDA_SHARE_MAP_TSS_NODES && map != SDA_SHARE_MAP_SYSTEMATIC) return fail(SDA_ERR_INVALID_ARGUMENT, "unknown share map %d", map);
    if (map == SDA_SHARE_MAP_SYSTEMATIC && !g->sys_default)
        return fail(SDA_ERR_UNSUPPORTED, "the systematic share map exists for packed Shamir with privacy_threshold > 0 on the "
                                         "matrix-form kernels (not the transform kernel, not additive sharing)");
    g->sys = map == SDA_SHARE_MAP_SYSTEMATIC;
    return SDA_OK;
}

// one call of the generator under the CSPRNG key `key` (the call key, or the stream key in deterministic mode)
static int generate_batch_impl(sda_share_generator_t* g, const DrbgKey& key, const int64_t* d_secrets,
                               size_t participants, size_t len, size_t secrets_stride,
                               const int64_t* d_rand, size_t rand_stride,
                               uint64_t first_participant, int64_t* d_out,
                               size_t out_stride_participant, size_t out_stride_clerk,
                               void* stream) {
    if (participants == 0 || len == 0) return SDA_OK;
    if (!d_secrets || !d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (!d_rand) SDA_TRY(check_streams(first_participant, participants));
    SDA_TRY(g->ctx.use());
    hipStream_t s = g->ctx.pick(stream);
    GenLayout L;
    L.secrets = d_secrets; L.secrets_stride = secrets_stride;
    L.rand = d_rand; L.rand_stride = rand_stride;
    L.out = d_out; L.out_stride_participant = out_stride_participant; L.out_stride_clerk = out_stride_clerk;
    L.participants = participants; L.len = len; L.first_participant = first_participant;
    // the device CSPRNG's draws are shares 0..t-1 themselves (systematic share map); injected randomness keeps tss's map
    const bool sys = !d_rand && effective_sys(g);
    L.direct_rows = sys ? g->t : 0;
    if (g->additive && g->rust_signed) {
        // the reference's representatives: shares 0..n-2 are the draws themselves, the last one the fold of (acc - r) % q
        // (additive.rs:42-47).  Draws that are not injected come from the same sda-drbg-v1 streams the canonical kernel uses,
        // inside the kernel: no scratch
        HIP_TRY(launch_additive_generate_signed(L, g->n, g->mod, key, g->drbg.rounds, s));
        return SDA_OK;
    }
    if (g->additive) {
        HIP_TRY(launch_additive_generate(L, g->n, g->mod, key, g->drbg.rounds, s));
        return SDA_OK;
    }
    switch (path_for_call(g->path, d_rand != nullptr, g->drbg.rounds)) {
        case FAM_N31:
            HIP_TRY(launch_packed_generate_n31(L, g->n, g->k, g->t, g->mod, g->n31p, sys ? *g->matarg_n31_sys : *g->matarg_n31, key, s));
            return SDA_OK;
        case FAM_NGEMM:
            HIP_TRY(launch_packed_generate_ngemm(L, g->mod, key, sys ? g->gplan_sys : g->gplan, s));
            return SDA_OK;
        case FAM_L31:
            HIP_TRY(launch_packed_generate_l31(L, g->n, g->k, g->t, g->mod, g->lp, sys ? *g->matarg_sys : *g->matarg, key, g->drbg.rounds, s));
            return SDA_OK;
        case FAM_MONT64:
            HIP_TRY(launch_packed_generate(L, g->n, g->k, g->t, g->mod, g->mont, sys ? *g->matarg_sys : *g->matarg, key, g->drbg.rounds, s));
            return SDA_OK;
        case FAM_L31_GLOBAL:
            HIP_TRY(launch_packed_generate_l31_global(L, g->n, g->k, g->t, g->mod, g->lp, (sys ? g->d_Msys : g->d_M).as<uint64_t>(), key,
                                                      g->drbg.rounds, s));
            return SDA_OK;
        case FAM_FFT:                                         // tss's own map by construction (effective_sys() is false here)
            HIP_TRY(launch_packed_generate_fft(L, g->mod, key, g->fplan, g->drbg.rounds, s));
            return SDA_OK;
        case FAM_MFMA:
            HIP_TRY(launch_packed_generate_mfma(L, g->n, g->k, g->t, g->mod, g->mont, (sys ? g->d_Msys : g->d_M).as<uint64_t>(), key,
                                                g->drbg.rounds, s));
            return SDA_OK;
        default: break;
    }
    // any-shape path: materialise the CSPRNG draws first (identical values to the fused path)
    if (!d_rand && g->t > 0) {
        const size_t batches = (len + g->k - 1) / g->k;
        size_t rstride = 0, total = 0;
        if (__builtin_mul_overflow(batches, (size_t)g->t, &rstride) || __builtin_mul_overflow(participants, rstride, &total) || total > (SIZE_MAX >> 3))
            return fail(SDA_ERR_INVALID_ARGUMENT, "participants * batches * privacy_threshold draws do not fit a buffer");
        SDA_TRY(g->d_rand.reserve(total * 8));
        HIP_TRY(launch_drbg_fill(g->d_rand.as<int64_t>(), rstride, participants, batches, g->t, first_participant, g->mod,
                                 key, g->drbg.rounds, s));
        L.rand = g->d_rand.as<int64_t>();
        L.rand_stride = rstride;
    }
    HIP_TRY(launch_packed_generate_generic(L, g->n, g->k, g->t, g->mod, g->mont, (sys ? g->d_Msys : g->d_M).as<uint64_t>(), s));
    return SDA_OK;
}

extern "C" int sda_share_generator_generate_batch_dev(sda_share_generator_t* g, const int64_t* d_secrets,
                                                      size_t participants, size_t len, size_t secrets_stride,
                                                      const int64_t* d_rand, size_t rand_stride,
                                                      uint64_t first_participant, int64_t* d_out,
                                                      size_t out_stride_participant, size_t out_stride_clerk,
                                                      void* stream) {
    if (!g) return fail(SDA_ERR_INVALID_ARGUMENT, "generator is NULL");
    if (participants == 0 || len == 0) return SDA_OK;
    DrbgKey key = d_rand ? DrbgKey{} : g->drbg.call_key();      // injected randomness draws nothing from the CSPRNG
    g_last_gen_kernel[0] = 0;
    const int st = generate_batch_impl(g, key, d_secrets, participants, len, secrets_stride, d_rand, rand_stride, first_participant,
                                       d_out, out_stride_participant, out_stride_clerk, stream);
    snprintf(g_last_call_kernels, sizeof g_last_call_kernels, "%s", g_last_gen_kernel);
    explicit_bzero(&key, sizeof key);
    return st;
}

extern "C" int sda_share_generator_generate(sda_share_generator_t* g, const int64_t* secrets, size_t len,
                                            const int64_t* rand, size_t rand_len, int64_t* out, size_t out_len) {
    if (!g) return fail(SDA_ERR_INVALID_ARGUMENT, "generator is NULL");
    const size_t B = (size_t)sda_share_generator_batch_count(g, len);
    if (out_len != (size_t)g->n * B) return fail(SDA_ERR_INVALID_ARGUMENT, "out_len must be share_count * batches = %zu (got %zu)", (size_t)g->n * B, out_len);
    if (len == 0) return SDA_OK;                    // zero batches: n empty vectors (batched.rs:25-28)
    if (!secrets || !out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    const size_t want_rand = B * g->t;
    if (rand && rand_len != want_rand)
        return fail(SDA_ERR_INVALID_ARGUMENT, "rand_len must be %zu (got %zu)", want_rand, rand_len);
    SDA_TRY(g->ctx.use());
    SDA_TRY(g->d_secrets.reserve(len * 8));
    const size_t Bs = B + (B & 1);                  // even clerk stride keeps 16-byte stores legal
    SDA_TRY(g->d_out.reserve((size_t)g->n * Bs * 8));
    HIP_TRY(hipMemcpyAsync(g->d_secrets.p, secrets, len * 8, hipMemcpyHostToDevice, g->ctx.stream));
    const int64_t* d_rand = nullptr;
    if (rand && want_rand) {                        // injected draws are staged in the handle's own scratch
        SDA_TRY(g->d_rand.reserve(want_rand * 8));
        HIP_TRY(hipMemcpyAsync(g->d_rand.p, rand, want_rand * 8, hipMemcpyHostToDevice, g->ctx.stream));
        d_rand = g->d_rand.as<int64_t>();
    }
    DrbgKey key = d_rand ? DrbgKey{} : g->drbg.call_key();
    const uint64_t stream_id = d_rand ? 0 : g->drbg.host_stream();
    g_last_gen_kernel[0] = 0;
    int st = generate_batch_impl(g, key, g->d_secrets.as<int64_t>(), 1, len, len, d_rand, want_rand, stream_id,
                                 g->d_out.as<int64_t>(), (size_t)g->n * Bs, Bs, nullptr);
    snprintf(g_last_call_kernels, sizeof g_last_call_kernels, "%s", g_last_gen_kernel);
    explicit_bzero(&key, sizeof key);
    if (st == SDA_OK) {
        hipError_t e = hipMemcpy2DAsync(out, B * 8, g->d_out.p, Bs * 8, B * 8, g->n, hipMemcpyDeviceToHost, g->ctx.stream);
        if (e != hipSuccess) st = fail(SDA_ERR_HIP, "hipMemcpy2DAsync failed: %s", hipGetErrorString(e));
    }
    if (st == SDA_OK) st = g->ctx.sync(); else (void)hipStreamSynchronize(g->ctx.stream);
    return st;
}

// =================================================================================================
// ShareCombiner
// =================================================================================================
struct sda_share_combiner {
    ModParams mod;
    Ctx ctx;
    AccState acc;
    DevBuf tile, d_out;
    size_t jobs = 0, dimension = 0;
    bool begun = false;
    unsigned max_wg_per_cu = 0;          // 0 = no cap (sda_share_combiner_set_residency)
    // Cross-stream ordering of the accumulators.  Every *_dev call that touches them records `ev_done` on its stream when it has
    // queued its work, and first makes its stream wait for the previous call's event when that call ran on ANOTHER stream.  So
    // generate_combine_dev on stream A followed by finish_dev on stream B is ordered by the library, not by the caller
    // (clerk.rs:80-86 has one thread and no streams: the reference's caller cannot be asked to keep such a rule).
    hipEvent_t ev_done = nullptr;
    bool pending = false;
    // (the wait is issued whenever work is pending, also on the stream that recorded it - there it costs nothing - instead of
    // comparing hipStream_t handles: a stream the caller destroyed and a new one that reuses its handle must not look alike)
    int join_pending(hipStream_t s) {
        if (pending) HIP_TRY(hipStreamWaitEvent(s, ev_done, 0));
        return SDA_OK;
    }
    int mark_pending(hipStream_t s) {
        if (!ev_done) HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_done, s));
        pending = true;
        return SDA_OK;
    }
};

extern "C" int sda_share_combiner_new(const sda_sharing_scheme_t* scheme, sda_share_combiner_t** out) {
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    SDA_TRY(check_scheme_kind(scheme));
    sda_share_combiner* c = new (std::nothrow) sda_share_combiner();
    if (!c) return fail(SDA_ERR_ALLOC, "out of memory");
    int st = make_mod(scheme->modulus, c->mod);      // sharing/mod.rs:62-68: both variants use the modulus only
    if (st == SDA_OK) st = c->ctx.init();
    if (st != SDA_OK) { sda_share_combiner_free(c); return st; }
    *out = c;
    return SDA_OK;
}

extern "C" void sda_share_combiner_free(sda_share_combiner_t* c) {
    if (!c) return;
    if (c->ctx.device >= 0) (void)hipSetDevice(c->ctx.device);
    c->acc.release(); c->tile.release(); c->d_out.release();
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    c->ctx.destroy();
    delete c;
}

extern "C" int sda_share_combiner_combine(sda_share_combiner_t* c, const int64_t* const* rows, const size_t* row_lens,
                                          size_t n_rows, int64_t* out, size_t out_cap, size_t* out_len) {
    if (!c || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (n_rows == 0) return SDA_OK;                                        // combiner.rs:17
    if (!rows || !row_lens) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL rows");
    const size_t dimension = row_lens[0];
    for (size_t r = 0; r < n_rows; ++r)
        if (row_lens[r] != dimension) return fail(SDA_ERR_WRONG_DIMENSION, "Wrong dimension");   // combiner.rs:21
    if (dimension == 0) return SDA_OK;
    if (!out || out_cap < dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small");
    for (size_t r = 0; r < n_rows; ++r)
        if (!rows[r]) return fail(SDA_ERR_INVALID_ARGUMENT, "rows[%zu] is NULL", r);
    SDA_TRY(column_sum_host(c->ctx, c->acc, c->tile, c->d_out, c->mod, rows, n_rows, dimension, out));
    *out_len = dimension;
    return SDA_OK;
}

extern "C" int sda_share_combiner_begin_dev(sda_share_combiner_t* c, size_t jobs, size_t dimension, void* stream) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (jobs > 65535) return fail(SDA_ERR_UNSUPPORTED, "at most 65535 jobs");
    SDA_TRY(c->ctx.use());
    SDA_TRY(c->join_pending(c->ctx.pick(stream)));
    SDA_TRY(c->acc.reset(jobs * dimension, c->ctx.pick(stream)));
    c->jobs = jobs; c->dimension = dimension; c->begun = true;
    return c->mark_pending(c->ctx.pick(stream));
}

extern "C" int sda_share_combiner_set_value_mode(sda_share_combiner_t* c, int mode) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    SDA_TRY(check_value_mode(mode));
    if (c->acc.rust_signed == (mode == SDA_VALUES_RUST_SIGNED)) return SDA_OK;   // unchanged: a running sum stays valid
    c->begun = false;                                               // the running sums are kept in ONE representation: begin again
    c->acc.rust_signed = mode == SDA_VALUES_RUST_SIGNED;            // combiner.rs:20-26 is the same loop for both sharing schemes
    c->acc.q = (int64_t)c->mod.m;
    return SDA_OK;
}

extern "C" int sda_share_combiner_update_dev(sda_share_combiner_t* c, const int64_t* d_shares, size_t job_stride,
                                             size_t n_rows, size_t row_stride, void* stream) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (!c->begun) return fail(SDA_ERR_STATE, "update before begin");
    if (n_rows == 0 || c->jobs == 0 || c->dimension == 0) return SDA_OK;
    if (!d_shares) return fail(SDA_ERR_INVALID_ARGUMENT, "d_shares is NULL");
    SDA_TRY(c->ctx.use());
    SDA_TRY(c->join_pending(c->ctx.pick(stream)));
    SDA_TRY(acc_update(c->acc, d_shares, c->jobs, job_stride, n_rows, row_stride, c->dimension, c->ctx.pick(stream), c->max_wg_per_cu));
    return c->mark_pending(c->ctx.pick(stream));
}

// software-pipelined step: tile i+1 is generated while tile i is summed, in one dual-role launch
extern "C" int sda_share_generator_generate_combine_dev(sda_share_generator_t* g, sda_share_combiner_t* c,
                                                        const int64_t* d_secrets, size_t participants, size_t len,
                                                        size_t secrets_stride, uint64_t first_participant,
                                                        int64_t* d_out, size_t out_stride_participant,
                                                        size_t out_stride_clerk, const int64_t* d_prev,
                                                        size_t prev_participants, void* stream) {
    if (!g || !c) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!c->begun) return fail(SDA_ERR_STATE, "combiner: update before begin");
    if (g->rust_signed || c->acc.rust_signed)
        return fail(SDA_ERR_UNSUPPORTED, "the dual-role launch sums in 128 bits: SDA_VALUES_RUST_SIGNED takes generate_batch_dev + update_dev");
    const size_t B = (size_t)sda_share_generator_batch_count(g, len);
    if (c->jobs != g->n || c->dimension != B)
        return fail(SDA_ERR_INVALID_ARGUMENT, "combiner must have been begun with jobs = share_count (%u) and dimension = batches (%zu)", g->n, B);
    if (participants > 0 && (!d_secrets || !d_out)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (prev_participants > 0 && !d_prev) return fail(SDA_ERR_INVALID_ARGUMENT, "d_prev is NULL");
    SDA_TRY(g->ctx.use());
    hipStream_t s = g->ctx.pick(stream);
    GenLayout L;
    L.secrets = d_secrets; L.secrets_stride = secrets_stride; L.rand = nullptr; L.rand_stride = 0;
    L.out = d_out; L.out_stride_participant = out_stride_participant; L.out_stride_clerk = out_stride_clerk;
    L.participants = len ? participants : 0; L.len = len; L.first_participant = first_participant;
    const bool sys = effective_sys(g);                                        // the dual-role launch always draws on the device
    L.direct_rows = sys ? g->t : 0;
    if (L.participants) SDA_TRY(check_streams(first_participant, participants));
    // the clerk sums this call adds may still be running on another stream when the caller next touches the combiner
    // (update_dev / finish_dev on a stream of its own): the combiner carries the event they wait for
    SDA_TRY(c->join_pending(s));
    DrbgKey key = g->drbg.call_key();
    bool fused = false;
    hipError_t he = hipSuccess;
    g_last_gen_kernel[0] = 0;
    if (g->additive) {
        he = launch_fused_additive(L, g->n, g->mod, key, g->drbg.rounds, c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(),
                                   d_prev, prev_participants, c->jobs, c->dimension, s, &fused);
    } else switch (fused_for_call(g->path, g->drbg.rounds)) {
        case FAM_N31:
            he = launch_fused_packed_n31(L, g->n, g->k, g->t, g->mod, g->n31p, sys ? *g->matarg_n31_sys : *g->matarg_n31, key,
                                         c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(), d_prev, prev_participants, c->jobs, c->dimension,
                                         s, &fused);
            break;
        case FAM_NGEMM:
        {
            const NGemmPlan& gp = sys ? g->gplan_sys : g->gplan;
            // the clerk waves' progress slots (grow-only; A/B knob SDA_NGEMM_CLERK_WG: rounds 4 - 5's clerk workgroups instead)
            const uint64_t slots = g->knob_ngemm_clerk_wg ? 0 : ngemm_clerk_slots(L, gp);
            if (slots && slots <= (1ull << 32)) {
                const int rs = g->d_cw_progress.reserve((size_t)slots * 16);
                if (rs != SDA_OK) { explicit_bzero(&key, sizeof key); return rs; }
            }
            he = launch_fused_packed_ngemm(L, g->mod, key, gp, c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(), d_prev,
                                           prev_participants, c->jobs, c->dimension, s, &fused,
                                           slots && slots <= (1ull << 32) ? g->d_cw_progress.as<uint64_t>() : nullptr, (size_t)slots);
            break;
        }
        case FAM_L31:
            he = launch_fused_packed_l31(L, g->n, g->k, g->t, g->mod, g->lp, sys ? *g->matarg_sys : *g->matarg, key, g->drbg.rounds,
                                         c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(), d_prev, prev_participants, c->jobs,
                                         c->dimension, s, &fused);
            break;
        case FAM_MFMA:
            he = launch_fused_packed_mfma(L, g->n, g->k, g->t, g->mod, g->mont, (sys ? g->d_Msys : g->d_M).as<uint64_t>(), key, g->drbg.rounds,
                                          c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(), d_prev, prev_participants, c->jobs,
                                          c->dimension, s, &fused);
            break;
        default: break;                                                       // no dual-role kernel for this family
    }
    int st = SDA_OK;
    if (he != hipSuccess) st = fail(SDA_ERR_HIP, "dual-role launch failed: %s", hipGetErrorString(he));
    // layouts / shapes the dual-role kernel does not cover: the two ordinary launches, same result.  The transform kernel is
    // bound by its vector ALUs (two workgroups per CU, LDS-limited, 0.9 TB/s of HBM traffic) and the clerk sum by HBM, so for
    // that shape the clerk sum of the previous tile is issued on a side stream and runs in the wave slots the transform
    // kernel leaves free: fork after whatever precedes this call on `stream`, join before anything that follows it
    const bool both = prev_participants > 0 && participants > 0 && len > 0;
    // (the narrow limb GEMM fills every SIMD's registers with its own nine waves per CU: a clerk sum on a side stream found no
    // room beside it - 18.9 ms per 500-participant tile of PSS_155_728_100 against 16.9 for the two launches back to back)
    const bool gemm_form = g->fft && g->ngemm && g->drbg.rounds == 20;
    // (round 6: single-wave clerk workgroups, which WOULD fit beside it - two per SIMD on the three SIMDs without the loader wave -
    // are not placed there either: 14.9 ms against 12.8 for the dual-role grid, profiles/r06/ab_ngemm_side_waves_not_adopted.txt)
    bool side = false;
    if (st == SDA_OK && !fused && both && g->fft && !gemm_form && !g->knob_no_side_stream) {
        side = true;
        st = g->side_stream();
        if (st == SDA_OK) {
            hipError_t e = hipEventRecord(g->ev_fork, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(g->aux, g->ev_fork, 0);
            if (e != hipSuccess) st = fail(SDA_ERR_HIP, "fork to the side stream failed: %s", hipGetErrorString(e));
        }
        if (st == SDA_OK)
            st = generate_batch_impl(g, key, d_secrets, participants, len, secrets_stride, nullptr, 0, first_participant,
                                     d_out, out_stride_participant, out_stride_clerk, stream);
        if (st == SDA_OK) {
            // one clerk-sum workgroup per CU (4 waves, 72 registers: what two transform workgroups leave free on every SIMD),
            // walking the job in grid strides
            const long ww = g->knob_side_wgs;                            // A/B only
            const unsigned walk = ww > 0 ? (unsigned)ww : 256u;
            hipError_t e = launch_combine_update(c->acc.lo.as<uint64_t>(), c->acc.hi.as<int64_t>(), d_prev, c->jobs, out_stride_clerk,
                                                 prev_participants, out_stride_participant, c->dimension, g->aux, 0, walk);
            if (e != hipSuccess) st = fail(SDA_ERR_HIP, "clerk-sum launch on the side stream failed: %s", hipGetErrorString(e));
        }
        // join even after a failure, so that `stream` never runs ahead of work already queued on the side stream (nothing to
        // join when the side stream itself could not be set up)
        if (g->aux && g->ev_join) {
            hipError_t e = hipEventRecord(g->ev_join, g->aux);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, g->ev_join, 0);
            if (e != hipSuccess && st == SDA_OK) st = fail(SDA_ERR_HIP, "join from the side stream failed: %s", hipGetErrorString(e));
        }
    } else {
        if (st == SDA_OK && !fused && prev_participants > 0)
            st = sda_share_combiner_update_dev(c, d_prev, out_stride_clerk, prev_participants, out_stride_participant, stream);
        if (st == SDA_OK && !fused && participants > 0 && len > 0)
            st = generate_batch_impl(g, key, d_secrets, participants, len, secrets_stride, nullptr, 0, first_participant,
                                     d_out, out_stride_participant, out_stride_clerk, stream);
    }
    explicit_bzero(&key, sizeof key);
    // every path above leaves clerk-sum work for `c` in flight on `s` (after the join): whoever next uses the combiner on ANOTHER
    // stream waits for this point (sda_share_combiner::join_pending)
    // - ALSO when the call failed part-way (e.g. the share-generation launch after the clerk-sum update was queued): a later
    // finish_dev on another stream must wait for what is already in flight
    if (prev_participants > 0) {
        const int mp = c->mark_pending(s);
        if (st == SDA_OK) st = mp;
    }
    if (fused) snprintf(g_last_call_kernels, sizeof g_last_call_kernels, "%s", g_last_gen_kernel);
    else snprintf(g_last_call_kernels, sizeof g_last_call_kernels, "%s%s%s", g_last_gen_kernel, g_last_gen_kernel[0] && prev_participants ? " + " : "",
                  prev_participants ? (side ? "combine_update_walk_kernel (side stream)" : "combine_update_kernel (two launches)") : "");
    return st;
}

extern "C" int sda_share_combiner_set_residency(sda_share_combiner_t* c, unsigned max_workgroups_per_cu) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (max_workgroups_per_cu > 8) return fail(SDA_ERR_INVALID_ARGUMENT, "max_workgroups_per_cu must be 0 (no cap) .. 8");
    c->max_wg_per_cu = max_workgroups_per_cu;
    return SDA_OK;
}

extern "C" int sda_share_combiner_finish_dev(sda_share_combiner_t* c, int64_t* d_out, void* stream) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (!c->begun) return fail(SDA_ERR_STATE, "finish before begin");
    if (c->jobs * c->dimension == 0) return SDA_OK;
    if (!d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "d_out is NULL");
    SDA_TRY(c->ctx.use());
    SDA_TRY(c->join_pending(c->ctx.pick(stream)));
    SDA_TRY(acc_finish(c->acc, c->jobs * c->dimension, c->mod, d_out, c->ctx.pick(stream)));
    return c->mark_pending(c->ctx.pick(stream));
}

extern "C" int sda_share_combiner_begin(sda_share_combiner_t* c, size_t dimension) {
    return sda_share_combiner_begin_dev(c, 1, dimension, nullptr);
}

extern "C" int sda_share_combiner_update(sda_share_combiner_t* c, const int64_t* shares, size_t n_rows, size_t row_stride) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (!c->begun) return fail(SDA_ERR_STATE, "update before begin");
    if (c->jobs != 1) return fail(SDA_ERR_STATE, "the host form feeds ONE job: the combiner was begun with %zu jobs (use update_dev)", c->jobs);
    if (n_rows == 0 || c->dimension == 0) return SDA_OK;
    if (!shares) return fail(SDA_ERR_INVALID_ARGUMENT, "shares is NULL");
    if (row_stride < c->dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "row_stride < dimension");
    std::vector<const int64_t*> rows(n_rows);
    for (size_t r = 0; r < n_rows; ++r) rows[r] = shares + r * row_stride;
    SDA_TRY(c->ctx.use());
    return accumulate_host_rows(c->ctx, c->acc, c->tile, rows.data(), n_rows, c->dimension);
}

extern "C" int sda_share_combiner_finish(sda_share_combiner_t* c, int64_t* out) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "combiner is NULL");
    if (!c->begun) return fail(SDA_ERR_STATE, "finish before begin");
    if (c->jobs != 1) return fail(SDA_ERR_STATE, "the host form returns ONE vector: the combiner was begun with %zu jobs (use finish_dev)", c->jobs);
    if (c->dimension == 0) return SDA_OK;
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    SDA_TRY(c->d_out.reserve(c->dimension * 8));
    SDA_TRY(sda_share_combiner_finish_dev(c, c->d_out.as<int64_t>(), nullptr));
    HIP_TRY(hipMemcpyAsync(out, c->d_out.p, c->dimension * 8, hipMemcpyDeviceToHost, c->ctx.stream));
    return c->ctx.sync();
}

extern "C" int sda_share_combiner_combine_dense(sda_share_combiner_t* c, const int64_t* shares, size_t n_rows,
                                                size_t dimension, size_t row_stride, int64_t* out) {
    SDA_TRY(sda_share_combiner_begin(c, dimension));
    SDA_TRY(sda_share_combiner_update(c, shares, n_rows, row_stride));
    return sda_share_combiner_finish(c, out);
}

// =================================================================================================
// SecretReconstructor
// =================================================================================================
struct sda_secret_reconstructor {
    sda_sharing_scheme_t scheme;
    bool additive = true;
    size_t dimension = 0;
    uint32_t n = 0, k = 1, t = 0;
    ModParams mod;
    MontParams mont{0, 0};
    Ctx ctx;
    AccState acc;
    DevBuf tile, d_out, d_R, d_R31, d_shares;
    std::vector<size_t> cached_indices;
    bool have_R = false;
    bool narrow = false;                 // p < 2^31: the one-limb reveal kernel (d_R31) when the shape and layout allow
    bool no_narrow = false;              // knob SDA_NO_NARROW when the handle was created (a live handle never switches kernels)
    N31Params n31p{};
};

extern "C" int sda_secret_reconstructor_new(const sda_sharing_scheme_t* scheme, size_t dimension,
                                            sda_secret_reconstructor_t** out) {
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    SDA_TRY(check_scheme_kind(scheme));
    sda_secret_reconstructor* r = new (std::nothrow) sda_secret_reconstructor();
    if (!r) return fail(SDA_ERR_ALLOC, "out of memory");
    r->scheme = *scheme;
    r->dimension = dimension;
    int st = make_mod(scheme->modulus, r->mod);
    if (st == SDA_OK) {
        if (sharing_is_additive(scheme)) {
            if (scheme->share_count < 1 || scheme->share_count > 65535)
                st = fail(SDA_ERR_INVALID_ARGUMENT, "share_count must be in 1..65535");
            r->additive = true;
            r->n = (uint32_t)scheme->share_count;
        } else {
            st = validate_packed(*scheme);
            if (st == SDA_OK) {
                r->additive = false;
                r->n = (uint32_t)scheme->share_count; r->k = (uint32_t)scheme->secret_count; r->t = (uint32_t)scheme->privacy_threshold;
                MontCtx mc = h_mont_ctx(r->mod.m);
                r->mont.p = mc.p; r->mont.pinv = mc.pinv;
            }
        }
    }
    if (st == SDA_OK) st = r->ctx.init();
    r->no_narrow = knob(KNOB_NO_NARROW) != 0;
    if (st != SDA_OK) { sda_secret_reconstructor_free(r); return st; }
    *out = r;
    return SDA_OK;
}

extern "C" void sda_secret_reconstructor_free(sda_secret_reconstructor_t* r) {
    if (!r) return;
    if (r->ctx.device >= 0) (void)hipSetDevice(r->ctx.device);
    r->acc.release(); r->tile.release(); r->d_out.release(); r->d_R.release(); r->d_R31.release(); r->d_shares.release();
    r->ctx.destroy();
    delete r;
}

extern "C" int sda_secret_reconstructor_set_value_mode(sda_secret_reconstructor_t* r, int mode) {
    if (!r) return fail(SDA_ERR_INVALID_ARGUMENT, "reconstructor is NULL");
    SDA_TRY(check_value_mode(mode));
    if (mode == SDA_VALUES_RUST_SIGNED && !r->additive) return fail(SDA_ERR_UNSUPPORTED, "%s", kNoSignedPacked);
    r->acc.rust_signed = mode == SDA_VALUES_RUST_SIGNED;
    r->acc.q = (int64_t)r->mod.m;
    return SDA_OK;
}

// k x n' Lagrange matrix of tss reconstruct: nodes {1} U {w3^(idx+1)}, evaluated at w2^e, e = 1..k;
// the column of node 1 (value 0) is dropped.  Identical for every batch, so it is built once per
// clerk-index set instead of once per batch (reference: Newton interpolation per batch).
static int prepare_R(sda_secret_reconstructor* r, const size_t* indices, size_t n_rows, hipStream_t s) {
    if (r->have_R && r->cached_indices.size() == n_rows && std::equal(indices, indices + n_rows, r->cached_indices.begin()))
        return SDA_OK;
    const uint64_t p = r->mod.m;
    const uint64_t w2 = h_canon(r->scheme.omega_secrets, p), w3 = h_canon(r->scheme.omega_shares, p);
    std::vector<uint64_t> nodes(n_rows + 1), evals(r->k);
    nodes[0] = 1;
    for (size_t c = 0; c < n_rows; ++c) nodes[c + 1] = h_powmod(w3, (uint64_t)indices[c] + 1, p);
    for (uint32_t e = 0; e < r->k; ++e) evals[e] = h_powmod(w2, e + 1, p);
    if (!h_all_distinct(nodes))
        return fail(SDA_ERR_INVALID_ARGUMENT, "clerk indices map to colliding evaluation points (duplicate index?)");
    std::vector<uint64_t> R;
    if (!h_lagrange_matrix_mont(nodes, evals, 0, p, R)) return fail(SDA_ERR_INVALID_ARGUMENT, "reconstruction matrix is singular");
    // every host-side step that can fail, and both reservations, come BEFORE the first asynchronous copy: R and R31 are
    // local vectors and must outlive the copies (which are synchronised below, on the only path that enqueues them)
    r->have_R = false;
    std::vector<int32_t> R31;
    r->narrow = p < (1ull << 31) && !r->no_narrow;
    SDA_TRY(r->d_R.reserve(R.size() * 8));
    if (r->narrow) {                                // the same matrix as centred int32 constants with R = 2^32
        uint64_t inv, inv32;
        if (!h_invmod(p, 1ull << 32, inv) || !h_invmod((1ull << 32) % p, p, inv32)) return fail(SDA_ERR_INVALID_ARGUMENT, "modulus not odd");
        r->n31p.p = (uint32_t)p; r->n31p.pinv = (uint32_t)((1ull << 32) - inv); r->n31p.h = (uint32_t)((p + 1) / 2); r->n31p.pad = 0;
        R31.resize(R.size());
        for (size_t i = 0; i < R.size(); ++i) {
            const uint64_t mr = h_mulmod(R[i], inv32, p);                    // R holds M * 2^64: M * 2^32 mod p
            R31[i] = mr > (p - 1) / 2 ? (int32_t)((int64_t)mr - (int64_t)p) : (int32_t)mr;
        }
        SDA_TRY(r->d_R31.reserve(R31.size() * 4 + 16));
    }
    hipError_t ce = hipMemcpyAsync(r->d_R.p, R.data(), R.size() * 8, hipMemcpyHostToDevice, s);
    if (ce == hipSuccess && r->narrow) ce = hipMemcpyAsync(r->d_R31.p, R31.data(), R31.size() * 4, hipMemcpyHostToDevice, s);
    const hipError_t se = hipStreamSynchronize(s);  // always: R, R31 are local vectors
    if (ce != hipSuccess || se != hipSuccess)
        return fail(SDA_ERR_HIP, "uploading the reconstruction matrix failed: %s", hipGetErrorString(ce != hipSuccess ? ce : se));
    r->cached_indices.assign(indices, indices + n_rows);
    r->have_R = true;
    return SDA_OK;
}

extern "C" int sda_secret_reconstructor_reconstruct_dev(sda_secret_reconstructor_t* r, const size_t* indices,
                                                        size_t n_rows, const int64_t* d_shares, size_t row_len,
                                                        size_t row_stride, int64_t* d_out, size_t out_cap,
                                                        size_t* out_len, void* stream) {
    if (!r || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    SDA_TRY(r->ctx.use());
    hipStream_t s = r->ctx.pick(stream);
    if (r->additive) {                                                       // additive.rs:55-73
        if (n_rows == 0 || row_len == 0) return SDA_OK;
        if (!d_shares || !d_out || out_cap < row_len) return fail(SDA_ERR_INVALID_ARGUMENT, "bad device buffers");
        SDA_TRY(r->acc.reset(row_len, s));
        SDA_TRY(acc_update(r->acc, d_shares, 1, 0, n_rows, row_stride, row_len, s));          // additive.rs:62-69
        SDA_TRY(acc_finish(r->acc, row_len, r->mod, d_out, s));
        *out_len = row_len;
        return SDA_OK;
    }
    if (n_rows < (size_t)r->t + r->k) return fail(SDA_ERR_NOT_ENOUGH_SHARES, "Not enough shares to reconstruct");   // packed_shamir.rs:75
    if (!indices) return fail(SDA_ERR_INVALID_ARGUMENT, "indices is NULL");
    const size_t batches = (r->dimension + r->k - 1) / r->k;                 // batched.rs:77
    if (batches == 0) return SDA_OK;
    if (row_len < batches) return fail(SDA_ERR_ASSERTION, "share vector shorter than the number of batches (index out of bounds, batched.rs:84)");
    if (!d_shares || !d_out || out_cap < r->dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "bad device buffers");
    if (n_rows > 0xFFFFFFFFull) return fail(SDA_ERR_UNSUPPORTED, "too many rows");
    SDA_TRY(prepare_R(r, indices, n_rows, s));
    if (r->narrow && packed_reconstruct_n31_available((uint32_t)n_rows, r->k, r->mod.m, d_shares, row_stride, d_out))
        HIP_TRY(launch_packed_reconstruct_n31(d_shares, row_stride, (uint32_t)n_rows, r->k, batches, r->dimension, r->mod, r->n31p,
                                              r->d_R31.as<int32_t>(), d_out, s));
    else
        HIP_TRY(launch_packed_reconstruct(d_shares, row_stride, (uint32_t)n_rows, r->k, batches, r->dimension, r->mod, r->mont,
                                          r->d_R.as<uint64_t>(), d_out, s));
    *out_len = r->dimension;
    return SDA_OK;
}

extern "C" int sda_secret_reconstructor_reconstruct(sda_secret_reconstructor_t* r, const size_t* indices,
                                                    const int64_t* const* rows, const size_t* row_lens, size_t n_rows,
                                                    int64_t* out, size_t out_cap, size_t* out_len) {
    if (!r || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (n_rows > 0 && (!rows || !row_lens)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL rows");
    if (r->additive) {
        if (n_rows == 0) return SDA_OK;                                       // additive.rs:57-60
        const size_t dimension = row_lens[0];
        for (size_t i = 0; i < n_rows; ++i)
            if (row_lens[i] != dimension) return fail(SDA_ERR_MISMATCHING_DIMENSION, "Mismatching dimension");   // additive.rs:64
        if (dimension == 0) return SDA_OK;
        if (!out || out_cap < dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small");
        SDA_TRY(column_sum_host(r->ctx, r->acc, r->tile, r->d_out, r->mod, rows, n_rows, dimension, out));
        *out_len = dimension;
        return SDA_OK;
    }
    if (n_rows < (size_t)r->t + r->k) return fail(SDA_ERR_NOT_ENOUGH_SHARES, "Not enough shares to reconstruct");
    const size_t batches = (r->dimension + r->k - 1) / r->k;
    if (batches == 0) return SDA_OK;
    for (size_t i = 0; i < n_rows; ++i)
        if (row_lens[i] < batches) return fail(SDA_ERR_ASSERTION, "share vector %zu shorter than the number of batches (index out of bounds, batched.rs:84)", i);
    if (!out || out_cap < r->dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small");
    SDA_TRY(r->ctx.use());
    const size_t stride = batches + (batches & 1);
    SDA_TRY(r->d_shares.reserve(n_rows * stride * 8));
    SDA_TRY(r->d_out.reserve(r->dimension * 8));
    for (size_t i = 0; i < n_rows; ++i)
        HIP_TRY(hipMemcpyAsync(r->d_shares.as<int64_t>() + i * stride, rows[i], batches * 8, hipMemcpyHostToDevice, r->ctx.stream));
    size_t n_out = 0;
    SDA_TRY(sda_secret_reconstructor_reconstruct_dev(r, indices, n_rows, r->d_shares.as<int64_t>(), batches, stride,
                                                     r->d_out.as<int64_t>(), r->dimension, &n_out, nullptr));
    HIP_TRY(hipMemcpyAsync(out, r->d_out.p, n_out * 8, hipMemcpyDeviceToHost, r->ctx.stream));
    SDA_TRY(r->ctx.sync());
    *out_len = n_out;
    return SDA_OK;
}

// =================================================================================================
// Masking
// =================================================================================================
namespace {
struct MaskCore {
    sda_masking_scheme_t scheme;
    ModParams mod{0, 0, 0, 0};
    Ctx ctx;
    AccState acc;
    DevBuf tile, d_a, d_b, d_out, d_seeds, d_flags, d_list;
    Drbg drbg;
    bool rust_signed = false;      // SDA_VALUES_RUST_SIGNED: full.rs:30,46-48,62 / chacha.rs:43,88 with Rust's own `%`
    int set_value_mode(int mode) {
        if (mode != SDA_VALUES_CANONICAL && mode != SDA_VALUES_RUST_SIGNED) return fail(SDA_ERR_INVALID_ARGUMENT, "unknown value mode %d", mode);
        rust_signed = mode == SDA_VALUES_RUST_SIGNED;
        return SDA_OK;
    }
    // (a + b) % q or (a - b) % q in the handle's representation
    int addsub(const int64_t* d_x, const int64_t* d_y, size_t len, bool subtract, int64_t* d_z, hipStream_t s) const {
        if (rust_signed) HIP_TRY(launch_addsub_signed(d_x, d_y, len, subtract, (int64_t)mod.m, d_z, s));
        else HIP_TRY(launch_addsub_mod(d_x, d_y, len, subtract, mod, d_z, s));
        return SDA_OK;
    }

    int init(const sda_masking_scheme_t* s) {
        if (!s) return fail(SDA_ERR_INVALID_ARGUMENT, "scheme is NULL");
        if (s->kind != SDA_MASKING_NONE && s->kind != SDA_MASKING_FULL && s->kind != SDA_MASKING_CHACHA)
            return fail(SDA_ERR_INVALID_ARGUMENT, "unknown masking scheme kind %d", s->kind);
        scheme = *s;
        if (s->kind != SDA_MASKING_NONE) {
            SDA_TRY(make_mod(s->modulus, mod));
            SDA_TRY(drbg.init());
            SDA_TRY(ctx.init());           // `None` is the identity and needs no device
        }
        return SDA_OK;
    }
    void destroy() {
        if (ctx.device >= 0) (void)hipSetDevice(ctx.device);
        acc.release(); tile.wipe_release(); d_a.wipe_release(); d_b.wipe_release(); d_out.wipe_release();
        d_seeds.wipe_release(); d_flags.release(); d_list.release();
        ctx.destroy();
        drbg.wipe();
    }
    size_t seed_words() const { return (size_t)((scheme.seed_bitsize + 31) / 32); }   // chacha.rs:31
};
}  // namespace

struct sda_secret_masker { MaskCore core; };
struct sda_mask_combiner { MaskCore core; };
struct sda_secret_unmasker { MaskCore core; };

template <typename H>
static int mask_handle_new(const sda_masking_scheme_t* scheme, H** out) {
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    H* h = new (std::nothrow) H();
    if (!h) return fail(SDA_ERR_ALLOC, "out of memory");
    int st = h->core.init(scheme);
    if (st != SDA_OK) { h->core.destroy(); delete h; return st; }
    *out = h;
    return SDA_OK;
}
template <typename H>
static void mask_handle_free(H* h) {
    if (!h) return;
    h->core.destroy();
    delete h;
}

extern "C" int sda_secret_masker_new(const sda_masking_scheme_t* s, sda_secret_masker_t** out) { return mask_handle_new(s, out); }
extern "C" void sda_secret_masker_free(sda_secret_masker_t* m) { mask_handle_free(m); }
extern "C" int sda_mask_combiner_new(const sda_masking_scheme_t* s, sda_mask_combiner_t** out) { return mask_handle_new(s, out); }
extern "C" void sda_mask_combiner_free(sda_mask_combiner_t* c) { mask_handle_free(c); }
extern "C" int sda_secret_unmasker_new(const sda_masking_scheme_t* s, sda_secret_unmasker_t** out) { return mask_handle_new(s, out); }
extern "C" void sda_secret_unmasker_free(sda_secret_unmasker_t* u) { mask_handle_free(u); }

extern "C" int sda_secret_masker_set_drbg_key(sda_secret_masker_t* m, const uint8_t key[32]) {
    if (!m || !key) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    m->core.drbg.set_stream_key(key);
    return SDA_OK;
}
extern "C" int sda_secret_masker_set_drbg_master_key(sda_secret_masker_t* m, const uint8_t key[32]) {
    if (!m || !key) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    m->core.drbg.set_master_key(key);
    return SDA_OK;
}
extern "C" int sda_secret_masker_set_drbg_rounds(sda_secret_masker_t* m, int rounds) {
    if (!m) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    return m->core.drbg.set_rounds(rounds);
}

extern "C" uint64_t sda_secret_masker_mask_len(const sda_secret_masker_t* m, size_t len) {
    if (!m) return 0;
    switch (m->core.scheme.kind) {
        case SDA_MASKING_FULL: return len;
        case SDA_MASKING_CHACHA: return m->core.seed_words();
        default: return 0;
    }
}

extern "C" int sda_secret_masker_mask(sda_secret_masker_t* m, const int64_t* secrets, size_t len, const int64_t* rand,
                                      size_t rand_len, int64_t* mask_out, size_t mask_cap, size_t* mask_len,
                                      int64_t* masked_out) {
    if (!m || !mask_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *mask_len = 0;
    if (len > 0 && (!secrets || !masked_out)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MaskCore& c = m->core;
    if (c.scheme.kind == SDA_MASKING_NONE) {                                  // none.rs:13-19
        if (len) memmove(masked_out, secrets, len * 8);
        return SDA_OK;
    }
    const size_t want_mask = (size_t)sda_secret_masker_mask_len(m, len);
    if (c.scheme.kind == SDA_MASKING_CHACHA && len != c.scheme.dimension)
        return fail(SDA_ERR_ASSERTION, "assertion failed: `(left == right)` (dimension %llu, secrets %zu) - chacha.rs:26",
                    (unsigned long long)c.scheme.dimension, len);
    if (want_mask > 0 && (!mask_out || mask_cap < want_mask)) return fail(SDA_ERR_INVALID_ARGUMENT, "mask buffer too small (need %zu)", want_mask);
    if (rand && rand_len != want_mask) return fail(SDA_ERR_INVALID_ARGUMENT, "rand_len must be %zu (got %zu)", want_mask, rand_len);
    SDA_TRY(c.ctx.use());
    hipStream_t s = c.ctx.stream;

    if (c.scheme.kind == SDA_MASKING_FULL) {                                  // full.rs:21-35
        if (len == 0) return SDA_OK;
        SDA_TRY(c.d_a.reserve(len * 8));
        SDA_TRY(c.d_b.reserve(len * 8));
        SDA_TRY(c.d_out.reserve(len * 8));
        HIP_TRY(hipMemcpyAsync(c.d_a.p, secrets, len * 8, hipMemcpyHostToDevice, s));
        if (rand && c.rust_signed) {
            // full.rs:25-30: the mask is the draws as they are, masked = (s + m) % q
            HIP_TRY(hipMemcpyAsync(c.d_b.p, rand, len * 8, hipMemcpyHostToDevice, s));
            SDA_TRY(c.addsub(c.d_a.as<int64_t>(), c.d_b.as<int64_t>(), len, false, c.d_out.as<int64_t>(), s));
        } else if (rand) {
            HIP_TRY(hipMemcpyAsync(c.d_b.p, rand, len * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(launch_addsub_mod(c.d_a.as<int64_t>(), c.d_b.as<int64_t>(), len, false, c.mod, c.d_out.as<int64_t>(), s));
            // canonical mask = (rand + 0) mod q
            SDA_TRY(c.tile.reserve(len * 8));
            HIP_TRY(hipMemsetAsync(c.d_a.p, 0, len * 8, s));
            HIP_TRY(launch_addsub_mod(c.d_b.as<int64_t>(), c.d_a.as<int64_t>(), len, false, c.mod, c.tile.as<int64_t>(), s));
            HIP_TRY(hipMemcpyAsync(c.d_b.p, c.tile.p, len * 8, hipMemcpyDeviceToDevice, s));
        } else {
            DrbgKey key = c.drbg.call_key();
            const hipError_t he = launch_full_mask_drbg(c.d_a.as<int64_t>(), len, 1, len, c.drbg.host_stream(), c.mod, key, c.drbg.rounds,
                                                        c.d_b.as<int64_t>(), len, c.d_out.as<int64_t>(), len, s);
            explicit_bzero(&key, sizeof key);
            HIP_TRY(he);
            if (c.rust_signed)      // the draws are in [0, q) either way; only the sum takes the reference's sign (secrets are unchecked i64)
                SDA_TRY(c.addsub(c.d_a.as<int64_t>(), c.d_b.as<int64_t>(), len, false, c.d_out.as<int64_t>(), s));
        }
        HIP_TRY(hipMemcpyAsync(mask_out, c.d_b.p, len * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(masked_out, c.d_out.p, len * 8, hipMemcpyDeviceToHost, s));
        SDA_TRY(c.ctx.sync());
        *mask_len = len;
        return SDA_OK;
    }

    // ChaCha - chacha.rs:24-54
    const size_t nw = c.seed_words();
    WipedVec<int64_t> seed(nw);
    if (rand) {
        for (size_t i = 0; i < nw; ++i) seed[i] = (int64_t)(uint32_t)(uint64_t)rand[i];    // words are u32 (chacha.rs:30-33)
    } else {
        WipedVec<uint32_t> raw(nw ? nw : 1);
        SDA_TRY(os_entropy(raw.data(), nw * 4));                              // OsRng seed, chacha.rs:29-33
        for (size_t i = 0; i < nw; ++i) seed[i] = (int64_t)raw[i];
    }
    WipedVec<uint32_t> key8(8);
    seed_to_key(seed.data(), nw, key8.data());
    if (len) {
        SDA_TRY(c.acc.reset(len, s));
        SDA_TRY(chacha_accumulate(c.ctx, key8, 1, len, c.mod, c.acc, c.d_seeds, c.d_flags, c.d_list));
        SDA_TRY(c.d_a.reserve(len * 8));
        SDA_TRY(c.d_b.reserve(len * 8));
        SDA_TRY(c.d_out.reserve(len * 8));
        HIP_TRY(launch_combine_finish(c.acc.lo.as<uint64_t>(), c.acc.hi.as<int64_t>(), len, c.mod, c.d_b.as<int64_t>(), s));
        HIP_TRY(hipMemcpyAsync(c.d_a.p, secrets, len * 8, hipMemcpyHostToDevice, s));
        SDA_TRY(c.addsub(c.d_a.as<int64_t>(), c.d_b.as<int64_t>(), len, false, c.d_out.as<int64_t>(), s));    // chacha.rs:41-44
        HIP_TRY(hipMemcpyAsync(masked_out, c.d_out.p, len * 8, hipMemcpyDeviceToHost, s));
        SDA_TRY(c.ctx.sync());
    }
    for (size_t i = 0; i < nw; ++i) mask_out[i] = seed[i];                    // chacha.rs:48-50
    *mask_len = nw;
    return SDA_OK;
}

// device-resident batch of participants (participate.rs:52-54 for a whole tile): Full and None schemes
extern "C" int sda_secret_masker_mask_batch_dev(sda_secret_masker_t* m, const int64_t* d_secrets, size_t participants,
                                                size_t len, size_t secrets_stride, uint64_t first_participant,
                                                int64_t* d_masks, size_t mask_stride, int64_t* d_masked,
                                                size_t masked_stride, void* stream) {
    if (!m) return fail(SDA_ERR_INVALID_ARGUMENT, "masker is NULL");
    MaskCore& c = m->core;
    if (c.rust_signed) return fail(SDA_ERR_UNSUPPORTED, "SDA_VALUES_RUST_SIGNED is served by the trait-shaped mask(); the batched device form emits canonical residues");
    if (participants == 0 || len == 0) return SDA_OK;
    if (!d_secrets || !d_masked) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (secrets_stride < len || masked_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "stride < len");
    if (c.scheme.kind == SDA_MASKING_CHACHA) {                                // chacha.rs:24-54 for every participant
        if (len != c.scheme.dimension)
            return fail(SDA_ERR_ASSERTION, "assertion failed: `(left == right)` (dimension %llu, secrets %zu) - chacha.rs:26",
                        (unsigned long long)c.scheme.dimension, len);
        const size_t nw = c.seed_words();
        if (!d_masks || mask_stride < nw) return fail(SDA_ERR_INVALID_ARGUMENT, "d_masks must hold %zu seed words per participant", nw);
        SDA_TRY(c.ctx.use());
        hipStream_t s = c.ctx.pick(stream);
        // one OS-entropy seed per participant (chacha.rs:29-33); the "mask" a participant sends is its seed (chacha.rs:48-50)
        WipedVec<uint32_t> raw(participants * nw ? participants * nw : 1);
        SDA_TRY(os_entropy(raw.data(), participants * nw * 4));
        WipedVec<int64_t> words(participants * nw);
        WipedVec<uint32_t> key8(participants * 8, 0u);
        for (size_t p = 0; p < participants; ++p)
            for (size_t i = 0; i < nw; ++i) {
                words[p * nw + i] = (int64_t)raw[p * nw + i];
                if (i < 8) key8[p * 8 + i] = raw[p * nw + i];
            }
        HIP_TRY(hipMemcpy2DAsync(d_masks, mask_stride * 8, words.data(), nw * 8, nw * 8, participants, hipMemcpyHostToDevice, s));
        SDA_TRY(c.d_seeds.reserve(participants * 32));
        HIP_TRY(hipMemcpyAsync(c.d_seeds.p, key8.data(), participants * 32, hipMemcpyHostToDevice, s));
        const uint64_t zone = rand03_zone(c.mod.m);
        const double p_rej = (double)(UINT64_MAX - zone + 1) / 18446744073709551616.0;
        const uint32_t* seeds = c.d_seeds.as<uint32_t>();
        if (p_rej * (double)len > 1.0 || len >= 0xFFFFFFF0ull) {               // rejections are the rule: exact order for all
            HIP_TRY(launch_chacha_apply_repair(seeds, nullptr, 0, nullptr, participants, nullptr, len, c.mod, zone, d_secrets,
                                               secrets_stride, d_masked, masked_stride, s));
            HIP_TRY(hipStreamSynchronize(s));                                  // the host vectors above are in flight until here
            return SDA_OK;
        }
        SDA_TRY(c.d_flags.reserve(participants * sizeof(RejectRecord)));
        HIP_TRY(hipMemsetAsync(c.d_flags.p, 0, participants * sizeof(RejectRecord), s));
        HIP_TRY(launch_chacha_apply_fast(seeds, participants, len, c.mod, zone, d_secrets, secrets_stride, d_masked, masked_stride,
                                         c.d_flags.as<RejectRecord>(), s));
        std::vector<RejectRecord> rec(participants);
        HIP_TRY(hipMemcpyAsync(rec.data(), c.d_flags.p, participants * sizeof(RejectRecord), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<uint32_t> shift, exact;
        for (size_t p = 0; p < participants; ++p)
            if (rec[p].count) (rec[p].count <= 3 ? shift : exact).push_back((uint32_t)p);
        if (!shift.empty() || !exact.empty()) {
            SDA_TRY(c.d_list.reserve((shift.size() + exact.size()) * 4));
            uint32_t* dl = c.d_list.as<uint32_t>();
            if (!shift.empty()) HIP_TRY(hipMemcpyAsync(dl, shift.data(), shift.size() * 4, hipMemcpyHostToDevice, s));
            if (!exact.empty()) HIP_TRY(hipMemcpyAsync(dl + shift.size(), exact.data(), exact.size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(launch_chacha_apply_repair(seeds, dl, shift.size(), dl + shift.size(), exact.size(), c.d_flags.as<RejectRecord>(),
                                               len, c.mod, zone, d_secrets, secrets_stride, d_masked, masked_stride, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        return SDA_OK;
    }
    if (c.scheme.kind == SDA_MASKING_NONE) {                                  // none.rs:13-19: identity, no mask
        HIP_TRY(hipMemcpy2DAsync(d_masked, masked_stride * 8, d_secrets, secrets_stride * 8, len * 8, participants,
                                 hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
        return SDA_OK;
    }
    if (!d_masks) return fail(SDA_ERR_INVALID_ARGUMENT, "d_masks is NULL");
    if (mask_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "stride < len");
    SDA_TRY(check_streams(first_participant, participants));
    SDA_TRY(c.ctx.use());
    DrbgKey key = c.drbg.call_key();
    const hipError_t he = launch_full_mask_drbg(d_secrets, secrets_stride, participants, len, first_participant, c.mod, key,
                                                c.drbg.rounds, d_masks, mask_stride, d_masked, masked_stride, c.ctx.pick(stream));
    explicit_bzero(&key, sizeof key);
    HIP_TRY(he);
    return SDA_OK;
}

extern "C" int sda_mask_combiner_combine(sda_mask_combiner_t* mc, const int64_t* const* rows, const size_t* row_lens,
                                         size_t n_rows, int64_t* out, size_t out_cap, size_t* out_len) {
    if (!mc || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (n_rows > 0 && (!rows || !row_lens)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL rows");
    MaskCore& c = mc->core;
    if (c.scheme.kind == SDA_MASKING_NONE) {                                  // none.rs:21-26
        for (size_t r = 0; r < n_rows; ++r)
            if (row_lens[r] != 0) return fail(SDA_ERR_ASSERTION, "assertion failed: masks.iter().all(|mask| mask.len() == 0) - none.rs:23");
        return SDA_OK;
    }
    if (c.scheme.kind == SDA_MASKING_FULL) {                                  // full.rs:37-52
        if (n_rows == 0) return SDA_OK;
        const size_t dimension = row_lens[0];
        for (size_t r = 0; r < n_rows; ++r)
            if (row_lens[r] != dimension) return fail(SDA_ERR_ASSERTION, "assertion failed: `(left == right)` (mask length) - full.rs:43");
        if (dimension == 0) return SDA_OK;
        if (!out || out_cap < dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small");
        for (size_t r = 0; r < n_rows; ++r)
            if (!rows[r]) return fail(SDA_ERR_INVALID_ARGUMENT, "rows[%zu] is NULL", r);
        c.acc.rust_signed = c.rust_signed;                                    // full.rs:46-48 is the combiner's loop
        c.acc.q = (int64_t)c.mod.m;
        SDA_TRY(column_sum_host(c.ctx, c.acc, c.tile, c.d_out, c.mod, rows, n_rows, dimension, out));
        *out_len = dimension;
        return SDA_OK;
    }
    // ChaCha - chacha.rs:56-77: result has the CONFIGURED dimension (:58), also for zero seeds.  Every mask is >= 0 and the
    // running value starts at 0, so `result += m; result %= q` never leaves [0, q): both value modes give the same numbers
    c.acc.rust_signed = false;
    const size_t dimension = (size_t)c.scheme.dimension;
    if (dimension == 0) return SDA_OK;
    if (!out || out_cap < dimension) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small");
    WipedVec<uint32_t> key8(n_rows * 8);
    for (size_t r = 0; r < n_rows; ++r) {
        if (row_lens[r] > 0 && !rows[r]) return fail(SDA_ERR_INVALID_ARGUMENT, "rows[%zu] is NULL", r);
        seed_to_key(rows[r], row_lens[r], key8.data() + r * 8);
    }
    SDA_TRY(c.ctx.use());
    SDA_TRY(c.acc.reset(dimension, c.ctx.stream));
    SDA_TRY(chacha_accumulate(c.ctx, key8, n_rows, dimension, c.mod, c.acc, c.d_seeds, c.d_flags, c.d_list));
    SDA_TRY(c.d_out.reserve(dimension * 8));
    HIP_TRY(launch_combine_finish(c.acc.lo.as<uint64_t>(), c.acc.hi.as<int64_t>(), dimension, c.mod, c.d_out.as<int64_t>(), c.ctx.stream));
    HIP_TRY(hipMemcpyAsync(out, c.d_out.p, dimension * 8, hipMemcpyDeviceToHost, c.ctx.stream));
    SDA_TRY(c.ctx.sync());
    *out_len = dimension;
    return SDA_OK;
}

extern "C" int sda_secret_masker_set_value_mode(sda_secret_masker_t* m, int mode) {
    return m ? m->core.set_value_mode(mode) : fail(SDA_ERR_INVALID_ARGUMENT, "masker is NULL");
}
extern "C" int sda_mask_combiner_set_value_mode(sda_mask_combiner_t* c, int mode) {
    return c ? c->core.set_value_mode(mode) : fail(SDA_ERR_INVALID_ARGUMENT, "mask combiner is NULL");
}
extern "C" int sda_secret_unmasker_set_value_mode(sda_secret_unmasker_t* u, int mode) {
    return u ? u->core.set_value_mode(mode) : fail(SDA_ERR_INVALID_ARGUMENT, "unmasker is NULL");
}

extern "C" int sda_secret_unmasker_unmask(sda_secret_unmasker_t* u, const int64_t* mask, size_t mask_len,
                                          const int64_t* masked, size_t masked_len, int64_t* out) {
    if (!u) return fail(SDA_ERR_INVALID_ARGUMENT, "unmasker is NULL");
    MaskCore& c = u->core;
    if (c.scheme.kind == SDA_MASKING_NONE) {                                  // none.rs:28-33
        if (mask_len != 0) return fail(SDA_ERR_ASSERTION, "assertion failed: `(left == right)` (mask must be empty) - none.rs:30");
        if (masked_len && (!masked || !out)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
        if (masked_len) memmove(out, masked, masked_len * 8);
        return SDA_OK;
    }
    if (mask_len != masked_len)                                               // full.rs:58, chacha.rs:83
        return fail(SDA_ERR_ASSERTION, "assertion failed: `(left == right)` (mask %zu, masked secrets %zu)", mask_len, masked_len);
    if (masked_len == 0) return SDA_OK;
    if (!mask || !masked || !out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    SDA_TRY(c.ctx.use());
    hipStream_t s = c.ctx.stream;
    SDA_TRY(c.d_a.reserve(masked_len * 8));
    SDA_TRY(c.d_b.reserve(masked_len * 8));
    SDA_TRY(c.d_out.reserve(masked_len * 8));
    HIP_TRY(hipMemcpyAsync(c.d_a.p, masked, masked_len * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c.d_b.p, mask, masked_len * 8, hipMemcpyHostToDevice, s));
    SDA_TRY(c.addsub(c.d_a.as<int64_t>(), c.d_b.as<int64_t>(), masked_len, true, c.d_out.as<int64_t>(), s));   // (ms - m) % q
    HIP_TRY(hipMemcpyAsync(out, c.d_out.p, masked_len * 8, hipMemcpyDeviceToHost, s));
    return c.ctx.sync();
}

// (masked - mask) mod q on device-resident vectors (full.rs:54-67, chacha.rs:79-93 once the masks are combined)
extern "C" int sda_secret_unmasker_unmask_dev(sda_secret_unmasker_t* u, const int64_t* d_mask, const int64_t* d_masked,
                                              size_t len, int64_t* d_out, void* stream) {
    if (!u) return fail(SDA_ERR_INVALID_ARGUMENT, "unmasker is NULL");
    MaskCore& c = u->core;
    if (len == 0) return SDA_OK;
    if (!d_masked || !d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (c.scheme.kind == SDA_MASKING_NONE) {                                  // none.rs:28-33
        if (d_out != d_masked) HIP_TRY(hipMemcpyAsync(d_out, d_masked, len * 8, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
        return SDA_OK;
    }
    if (!d_mask) return fail(SDA_ERR_INVALID_ARGUMENT, "d_mask is NULL");
    SDA_TRY(c.ctx.use());
    return c.addsub(d_masked, d_mask, len, true, d_out, c.ctx.pick(stream));
}

extern "C" int sda_positive(const int64_t* values, size_t len, int64_t modulus, int64_t* out) {
    if (len && (!values || !out)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    // receive.rs:15 - `if v < 0 { v + modulus } else { v }`, no validation there either; formed in unsigned arithmetic so that a
    // hostile (value, modulus) pair wraps like release Rust instead of being undefined behaviour here
    for (size_t i = 0; i < len; ++i) out[i] = values[i] < 0 ? (int64_t)((uint64_t)values[i] + (uint64_t)modulus) : values[i];
    return SDA_OK;
}

// =================================================================================================
// Share-vector wire codec (zig-zag LEB128) - sodium.rs:36-41, :83-89
// =================================================================================================
struct sda_varint_codec {
    Ctx ctx;
    DevBuf d_blocks, d_offs, d_aux, d_total, d_status, d_in, d_out, d_rowoff;
};

extern "C" int sda_varint_codec_new(sda_varint_codec_t** out) {
    if (!out) return fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    sda_varint_codec* c = new (std::nothrow) sda_varint_codec();
    if (!c) return fail(SDA_ERR_ALLOC, "out of memory");
    int st = c->ctx.init();
    if (st == SDA_OK) st = c->d_total.reserve(8);
    if (st == SDA_OK) st = c->d_status.reserve(4);
    if (st != SDA_OK) { sda_varint_codec_free(c); return st; }
    *out = c;
    return SDA_OK;
}

extern "C" void sda_varint_codec_free(sda_varint_codec_t* c) {
    if (!c) return;
    if (c->ctx.device >= 0) (void)hipSetDevice(c->ctx.device);
    c->d_blocks.release(); c->d_offs.release(); c->d_aux.release(); c->d_total.release(); c->d_status.release();
    c->d_in.release(); c->d_out.release(); c->d_rowoff.release();
    delete c;
}

extern "C" size_t sda_varint_max_encoded_size(size_t count) { return count * 10; }

extern "C" int sda_varint_encode_dev(sda_varint_codec_t* c, const int64_t* d_values, size_t rows, size_t len,
                                     size_t row_stride, uint8_t* d_out, size_t out_cap, uint64_t* d_row_offsets,
                                     uint64_t* total_bytes, void* stream) {
    if (!c || !total_bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *total_bytes = 0;
    SDA_TRY(c->ctx.use());
    hipStream_t s = c->ctx.pick(stream);
    const size_t nb = varint_encode_blocks(rows, len);
    if (nb == 0) {
        if (d_row_offsets && rows + 1 > 0) HIP_TRY(hipMemsetAsync(d_row_offsets, 0, (rows + 1) * 8, s));
        return SDA_OK;
    }
    if (!d_values || !d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (row_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "row_stride < len");
    SDA_TRY(c->d_blocks.reserve(nb * 4));
    SDA_TRY(c->d_offs.reserve(nb * 8));
    SDA_TRY(c->d_aux.reserve(scan_aux_entries(nb) * 8));
    VarintRows R{d_values, rows, len, row_stride};
    HIP_TRY(launch_varint_lengths(R, c->d_blocks.as<uint32_t>(), s));
    HIP_TRY(launch_scan_u32(c->d_blocks.as<uint32_t>(), c->d_offs.as<uint64_t>(), nb, c->d_total.as<uint64_t>(),
                            c->d_aux.as<uint64_t>(), s));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, c->d_total.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (total > out_cap) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small: need %llu bytes", (unsigned long long)total);
    HIP_TRY(launch_varint_write(R, c->d_offs.as<uint64_t>(), d_out, d_row_offsets, s));
    if (d_row_offsets) HIP_TRY(hipMemcpyAsync(d_row_offsets + rows, c->d_total.p, 8, hipMemcpyDeviceToDevice, s));
    *total_bytes = total;
    return SDA_OK;
}

static int varint_count_scan(sda_varint_codec_t* c, const uint8_t* d_bytes, size_t n_bytes, hipStream_t s) {
    const size_t nb = varint_decode_blocks(n_bytes);
    SDA_TRY(c->d_blocks.reserve((nb + 1) * 4));
    SDA_TRY(c->d_offs.reserve((nb + 1) * 8));
    SDA_TRY(c->d_aux.reserve(scan_aux_entries(nb + 1) * 8));
    HIP_TRY(launch_varint_count(d_bytes, n_bytes, c->d_blocks.as<uint32_t>(), s));
    // one extra (zero) entry so that prefix(x) is defined for x == n_bytes on a block boundary
    HIP_TRY(hipMemsetAsync(c->d_blocks.as<uint32_t>() + nb, 0, 4, s));
    HIP_TRY(launch_scan_u32(c->d_blocks.as<uint32_t>(), c->d_offs.as<uint64_t>(), nb + 1, c->d_total.as<uint64_t>(),
                            c->d_aux.as<uint64_t>(), s));
    return SDA_OK;
}

// Row streaming needs a wave per row to keep the chip busy; fewer rows take the three-pass scan form,
// which parallelises inside a row.  SDA_VARINT_PATH=stream|scan pins one form (A/B runs, parity tests).
static bool varint_use_stream(size_t rows) {
    if (const long e = knob(KNOB_VARINT_PATH)) {
        if (e == 1) return rows > 0;
        if (e == 2) return false;
    }
    // measured (profiles/r01/wire_bench.json, rows of 349526 values): 1024 rows - decode equal, fused clerk sum 100
    // vs 123 Gvalues/s; 2000 rows - 229 vs 153 and 195 vs 125; 16000 rows - 260 vs 156 and 304 vs 127
    return rows >= 1536;
}

extern "C" int sda_varint_decode_dev(sda_varint_codec_t* c, const uint8_t* d_bytes, size_t n_bytes,
                                     const uint64_t* d_row_offsets, size_t rows, size_t len, int64_t* d_values,
                                     size_t row_stride, uint32_t* d_status, void* stream) {
    if (!c || !d_status) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!d_row_offsets && rows > 1) return fail(SDA_ERR_INVALID_ARGUMENT, "d_row_offsets is required for rows > 1");
    if (row_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "row_stride < len");
    SDA_TRY(c->ctx.use());
    hipStream_t s = c->ctx.pick(stream);
    if (n_bytes > 0 && (!d_bytes || !d_values)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (varint_use_stream(rows)) {       // many rows: one wave streams each row, the bytes are read once
        HIP_TRY(launch_varint_stream_decode(d_bytes, n_bytes, RowRanges{d_row_offsets, nullptr, 0}, rows, len, row_stride, d_values,
                                            d_status, s));
        return SDA_OK;
    }
    SDA_TRY(varint_count_scan(c, d_bytes, n_bytes, s));
    HIP_TRY(launch_varint_decode(d_bytes, n_bytes, c->d_offs.as<uint64_t>(), rows, len, row_stride, d_values, d_status, s));
    HIP_TRY(launch_varint_rowcheck(d_bytes, n_bytes, d_row_offsets, rows, len, c->d_offs.as<uint64_t>(), d_status, s));
    return SDA_OK;
}

extern "C" int sda_varint_encode(sda_varint_codec_t* c, const int64_t* values, size_t len, uint8_t* out, size_t out_cap,
                                 size_t* out_len) {
    if (!c || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (len == 0) return SDA_OK;
    if (!values || !out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    SDA_TRY(c->ctx.use());
    SDA_TRY(c->d_in.reserve(len * 8));
    SDA_TRY(c->d_out.reserve(len * 10 + 16));
    HIP_TRY(hipMemcpyAsync(c->d_in.p, values, len * 8, hipMemcpyHostToDevice, c->ctx.stream));
    uint64_t total = 0;
    SDA_TRY(sda_varint_encode_dev(c, c->d_in.as<int64_t>(), 1, len, len, c->d_out.as<uint8_t>(), len * 10, nullptr, &total, nullptr));
    if (total > out_cap) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small: need %llu bytes", (unsigned long long)total);
    HIP_TRY(hipMemcpyAsync(out, c->d_out.p, total, hipMemcpyDeviceToHost, c->ctx.stream));
    SDA_TRY(c->ctx.sync());
    *out_len = (size_t)total;
    return SDA_OK;
}

extern "C" int sda_varint_decode(sda_varint_codec_t* c, const uint8_t* bytes, size_t n_bytes, int64_t* out, size_t out_cap,
                                 size_t* out_len) {
    if (!c || !out_len) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (n_bytes == 0) return SDA_OK;
    if (!bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    SDA_TRY(c->ctx.use());
    hipStream_t s = c->ctx.stream;
    SDA_TRY(c->d_in.reserve(n_bytes + 16));
    HIP_TRY(hipMemcpyAsync(c->d_in.p, bytes, n_bytes, hipMemcpyHostToDevice, s));
    // first pass only to learn the value count
    SDA_TRY(varint_count_scan(c, c->d_in.as<uint8_t>(), n_bytes, s));
    uint64_t count = 0;
    HIP_TRY(hipMemcpyAsync(&count, c->d_total.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (count > out_cap) return fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small: %llu values", (unsigned long long)count);
    SDA_TRY(c->d_out.reserve((count ? count : 1) * 8));
    HIP_TRY(hipMemsetAsync(c->d_status.p, 0, 4, s));
    SDA_TRY(sda_varint_decode_dev(c, c->d_in.as<uint8_t>(), n_bytes, nullptr, 1, count, c->d_out.as<int64_t>(), count,
                                  c->d_status.as<uint32_t>(), nullptr));
    uint32_t status = 0;
    HIP_TRY(hipMemcpyAsync(&status, c->d_status.p, 4, hipMemcpyDeviceToHost, s));
    if (count) HIP_TRY(hipMemcpyAsync(out, c->d_out.p, count * 8, hipMemcpyDeviceToHost, s));
    SDA_TRY(c->ctx.sync());
    if (status) return fail(SDA_ERR_INVALID_ARGUMENT, "malformed varint stream (status %u: 1 = over-long value, 4 = ends inside a value)", status);
    *out_len = (size_t)count;
    return SDA_OK;
}

// streaming clerk: wire-format vectors -> decode tile -> clerk-sum update -> discard (clerk.rs:71-86)
extern "C" int sda_share_combiner_update_varint_dev(sda_share_combiner_t* c, sda_varint_codec_t* codec,
                                                    const uint8_t* d_bytes, size_t n_bytes, const uint64_t* d_row_offsets,
                                                    size_t rows, uint32_t* d_status, void* stream) {
    if (!c || !codec) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!c->begun) return fail(SDA_ERR_STATE, "update before begin");
    if (c->acc.rust_signed) return fail(SDA_ERR_UNSUPPORTED, "the wire-fed updates sum in 128 bits: SDA_VALUES_RUST_SIGNED takes decode + update[_dev]");
    if (rows == 0 || c->jobs == 0) return SDA_OK;
    if (rows % c->jobs) return fail(SDA_ERR_INVALID_ARGUMENT, "rows (%zu) must be a multiple of the combiner's jobs (%zu): job-major rows", rows, c->jobs);
    if (!d_status) return fail(SDA_ERR_INVALID_ARGUMENT, "d_status is NULL");
    if (!d_row_offsets && rows > 1) return fail(SDA_ERR_INVALID_ARGUMENT, "d_row_offsets is required for rows > 1");
    if (n_bytes > 0 && !d_bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    const size_t L = c->dimension, stride = L + (L & 1), rows_per_job = rows / c->jobs;
    SDA_TRY(c->ctx.use());
    if (varint_use_stream(rows)) {
        // many rows: 16 rows of a job per workgroup, decoded values summed in an LDS column window - no decoded tile
        const RowRanges rr{d_row_offsets, nullptr, 0};
        SDA_TRY(c->join_pending(c->ctx.pick(stream)));
        HIP_TRY(launch_varint_stream_combine(d_bytes, n_bytes, rr, c->jobs, rows_per_job, L, c->acc.lo.as<uint64_t>(),
                                             c->acc.hi.as<int64_t>(), d_status, c->ctx.pick(stream)));
        if (L == 0) HIP_TRY(launch_varint_stream_decode(d_bytes, n_bytes, rr, rows, 0, 0, nullptr, d_status, c->ctx.pick(stream)));
        return c->mark_pending(c->ctx.pick(stream));
    }
    SDA_TRY(c->tile.reserve((rows * stride ? rows * stride : 1) * 8));
    SDA_TRY(sda_varint_decode_dev(codec, d_bytes, n_bytes, d_row_offsets, rows, L, c->tile.as<int64_t>(), stride, d_status, stream));
    if (L == 0) return SDA_OK;
    return sda_share_combiner_update_dev(c, c->tile.as<int64_t>(), rows_per_job * stride, rows_per_job, stride, stream);
}

// ---- slotted rows: row r at d_bytes + r*slot_bytes, d_row_bytes[r] bytes long (single-pass kernels only) ----------
extern "C" size_t sda_varint_slot_size(size_t len) { return (len * 10 + 15) / 16 * 16; }

static int check_slots(const void* d_bytes, size_t slot_bytes, size_t len) {
    if (slot_bytes % 16 || ((uintptr_t)d_bytes & 15u)) return fail(SDA_ERR_INVALID_ARGUMENT, "slots must be 16-byte aligned (buffer and slot_bytes)");
    if (slot_bytes < len * 10) return fail(SDA_ERR_INVALID_ARGUMENT, "slot_bytes < 10 * len (sda_varint_slot_size)");
    return SDA_OK;
}

extern "C" int sda_varint_encode_rows_dev(sda_varint_codec_t* c, const int64_t* d_values, size_t rows, size_t len,
                                          size_t row_stride, uint8_t* d_out, size_t slot_bytes, uint64_t* d_row_bytes,
                                          void* stream) {
    if (!c) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (rows == 0) return SDA_OK;
    if (!d_out || !d_row_bytes || (len > 0 && !d_values)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (row_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "row_stride < len");
    SDA_TRY(check_slots(d_out, slot_bytes, len));
    SDA_TRY(c->ctx.use());
    HIP_TRY(launch_varint_stream_encode(VarintRows{d_values, rows, len, row_stride}, d_out, slot_bytes, d_row_bytes, c->ctx.pick(stream)));
    return SDA_OK;
}

extern "C" int sda_varint_decode_rows_dev(sda_varint_codec_t* c, const uint8_t* d_bytes, size_t slot_bytes,
                                          const uint64_t* d_row_bytes, size_t rows, size_t len, int64_t* d_values,
                                          size_t row_stride, uint32_t* d_status, void* stream) {
    if (!c || !d_status) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (rows == 0) return SDA_OK;
    if (!d_bytes || !d_row_bytes || (len > 0 && !d_values)) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (row_stride < len) return fail(SDA_ERR_INVALID_ARGUMENT, "row_stride < len");
    if (slot_bytes % 16 || ((uintptr_t)d_bytes & 15u)) return fail(SDA_ERR_INVALID_ARGUMENT, "slots must be 16-byte aligned (buffer and slot_bytes)");
    SDA_TRY(c->ctx.use());
    HIP_TRY(launch_varint_stream_decode(d_bytes, rows * slot_bytes, RowRanges{nullptr, d_row_bytes, slot_bytes}, rows, len, row_stride,
                                        d_values, d_status, c->ctx.pick(stream)));
    return SDA_OK;
}

extern "C" int sda_share_combiner_update_varint_rows_dev(sda_share_combiner_t* c, sda_varint_codec_t* codec,
                                                         const uint8_t* d_bytes, size_t slot_bytes, const uint64_t* d_row_bytes,
                                                         size_t rows, uint32_t* d_status, void* stream) {
    if (!c || !codec) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!c->begun) return fail(SDA_ERR_STATE, "update before begin");
    if (c->acc.rust_signed) return fail(SDA_ERR_UNSUPPORTED, "the wire-fed updates sum in 128 bits: SDA_VALUES_RUST_SIGNED takes decode + update[_dev]");
    if (rows == 0 || c->jobs == 0) return SDA_OK;
    if (rows % c->jobs) return fail(SDA_ERR_INVALID_ARGUMENT, "rows (%zu) must be a multiple of the combiner's jobs (%zu): job-major rows", rows, c->jobs);
    if (!d_status || !d_bytes || !d_row_bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (slot_bytes % 16 || ((uintptr_t)d_bytes & 15u)) return fail(SDA_ERR_INVALID_ARGUMENT, "slots must be 16-byte aligned (buffer and slot_bytes)");
    SDA_TRY(c->ctx.use());
    const RowRanges rr{nullptr, d_row_bytes, slot_bytes};
    const size_t L = c->dimension;
    SDA_TRY(c->join_pending(c->ctx.pick(stream)));
    HIP_TRY(launch_varint_stream_combine(d_bytes, rows * slot_bytes, rr, c->jobs, rows / c->jobs, L, c->acc.lo.as<uint64_t>(),
                                         c->acc.hi.as<int64_t>(), d_status, c->ctx.pick(stream)));
    if (L == 0) HIP_TRY(launch_varint_stream_decode(d_bytes, rows * slot_bytes, rr, rows, 0, 0, nullptr, d_status, c->ctx.pick(stream)));
    return c->mark_pending(c->ctx.pick(stream));
}

extern "C" int sda_share_combiner_update_varint(sda_share_combiner_t* c, sda_varint_codec_t* codec, const uint8_t* bytes,
                                                size_t n_bytes) {
    if (!c || !codec) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!c->begun) return fail(SDA_ERR_STATE, "update before begin");
    if (c->acc.rust_signed) return fail(SDA_ERR_UNSUPPORTED, "the wire-fed updates sum in 128 bits: SDA_VALUES_RUST_SIGNED takes decode + update[_dev]");
    if (c->jobs != 1) return fail(SDA_ERR_STATE, "the host form takes one participant's vector for ONE job (begin with jobs == 1)");
    if (n_bytes > 0 && !bytes) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL buffer");
    SDA_TRY(c->ctx.use());
    hipStream_t s = c->ctx.stream;
    SDA_TRY(codec->d_in.reserve(n_bytes + 16));
    if (n_bytes) HIP_TRY(hipMemcpyAsync(codec->d_in.p, bytes, n_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(codec->d_status.p, 0, 4, s));
    // a tile that fails the check must not pollute the running sums: decode first, test, then add
    const size_t L = c->dimension, stride = L + (L & 1);
    SDA_TRY(c->tile.reserve((stride ? stride : 1) * 8));
    SDA_TRY(sda_varint_decode_dev(codec, codec->d_in.as<uint8_t>(), n_bytes, nullptr, 1, L, c->tile.as<int64_t>(), stride,
                                  codec->d_status.as<uint32_t>(), nullptr));
    uint32_t status = 0;
    HIP_TRY(hipMemcpyAsync(&status, codec->d_status.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (status & 2u) return fail(SDA_ERR_WRONG_DIMENSION, "Wrong dimension");                       // combiner.rs:21
    if (status) return fail(SDA_ERR_INVALID_ARGUMENT, "malformed varint stream (status %u)", status);
    if (L == 0) return SDA_OK;
    return sda_share_combiner_update_dev(c, c->tile.as<int64_t>(), 0, 1, stride, nullptr);
}

// =================================================================================================
// multi-GPU helper, synthetic input, timing
// =================================================================================================
static int device_ready() {
    if (sda_device_count() == 0) return fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    return SDA_OK;
}
int capi_device_ready() {
    SDA_TRY(device_ready());
    HIP_TRY(hipSetDevice(g_device < sda_device_count() ? g_device : 0));
    return SDA_OK;
}

extern "C" int sda_modsum_parts_dev(int64_t modulus, const int64_t* d_parts, size_t parts, size_t part_stride,
                                    size_t len, int64_t* d_out, void* stream) {
    ModParams mod;
    SDA_TRY(make_mod(modulus, mod));
    SDA_TRY(device_ready());
    if (len == 0) return SDA_OK;
    if (!d_parts || !d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    HIP_TRY(launch_modsum_parts(d_parts, parts, part_stride, len, mod, d_out, reinterpret_cast<hipStream_t>(stream)));
    return SDA_OK;
}

extern "C" int sda_fill_synthetic_dev(int64_t* d_out, size_t participants, size_t len, size_t stride,
                                      uint64_t first_participant, uint64_t seed, int64_t modulus, void* stream) {
    ModParams mod;
    SDA_TRY(make_mod(modulus, mod));
    SDA_TRY(device_ready());
    if (!d_out) return fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    HIP_TRY(launch_fill_synthetic(d_out, participants, len, stride, first_participant, seed, mod, reinterpret_cast<hipStream_t>(stream)));
    return SDA_OK;
}

extern "C" int sda_event_create(void** ev) {
    if (!ev) return fail(SDA_ERR_INVALID_ARGUMENT, "ev is NULL");
    SDA_TRY(device_ready());
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return SDA_OK;
}
extern "C" int sda_event_destroy(void* ev) {
    if (ev) HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return SDA_OK;
}
extern "C" int sda_event_record(void* ev, void* stream) {
    HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
    return SDA_OK;
}
extern "C" int sda_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return fail(SDA_ERR_INVALID_ARGUMENT, "ms is NULL");
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return SDA_OK;
}
