// C ABI of the batched sealed-box open / seal (include/sda_hip.h "Sealed boxes").
// Host logic only: argument checks mirroring the reference's failure ("Sodium decryption failure", sodium.rs:80),
// scratch management, OS entropy for the ephemeral keys, kernel launches (sealedbox_kernels.hip).  No CPU fallback.
#include "../../include/sda_hip.h"

#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <string.h>
#include <sys/random.h>

#include <new>
#include <vector>

#include "capi_internal.hpp"
#include "kernels.hpp"

using namespace sda;

struct sda_sealedbox {
    int device = 0;
    void* d_states = nullptr; size_t states_cap = 0;       // SboxState per row
    void* d_partial = nullptr; size_t partial_cap = 0;     // Poly1305 partial sums
    void* d_keys = nullptr; size_t keys_cap = 0;           // ephemeral secrets + recipient keys (seal)
    void* d_io = nullptr; size_t io_cap = 0;               // host-form staging
};

namespace {
int reserve(void*& p, size_t& cap, size_t bytes, bool wipe) {
    if (bytes <= cap) return SDA_OK;
    if (p) {
        if (wipe) (void)hipMemset(p, 0, cap);
        (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    if (hipMalloc(&p, bytes < 256 ? 256 : bytes) != hipSuccess) { p = nullptr; return capi_fail(SDA_ERR_ALLOC, "hipMalloc(%zu) failed", bytes); }
    cap = bytes < 256 ? 256 : bytes;
    return SDA_OK;
}
int scratch(sda_sealedbox* b, size_t rows, size_t max_msg) {
    if (int st = reserve(b->d_states, b->states_cap, rows * sizeof(SboxState), true)) return st;
    return reserve(b->d_partial, b->partial_cap, rows * sbox_regions(max_msg) * 5 * sizeof(uint32_t), true);
}
int entropy(void* buf, size_t len) {
    uint8_t* p = static_cast<uint8_t*>(buf);
    size_t got = 0;
    while (got < len) {
        ssize_t r = getrandom(p + got, len - got, 0);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return capi_fail(SDA_ERR_ENTROPY, "getrandom failed (errno %d)", errno);
        got += (size_t)r;
    }
    return SDA_OK;
}
bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
}  // namespace

extern "C" int sda_sealedbox_new(sda_sealedbox_t** out) {
    if (!out) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (int st = capi_device_ready()) return st;
    sda_sealedbox* b = new (std::nothrow) sda_sealedbox();
    if (!b) return capi_fail(SDA_ERR_ALLOC, "out of memory");
    if (hipGetDevice(&b->device) != hipSuccess) { delete b; return capi_fail(SDA_ERR_HIP, "hipGetDevice failed"); }
    *out = b;
    return SDA_OK;
}

extern "C" void sda_sealedbox_free(sda_sealedbox_t* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    void* bufs[4] = {b->d_states, b->d_partial, b->d_keys, b->d_io};
    size_t caps[4] = {b->states_cap, b->partial_cap, b->keys_cap, b->io_cap};
    for (int i = 0; i < 4; ++i)
        if (bufs[i]) { (void)hipMemset(bufs[i], 0, caps[i]); (void)hipFree(bufs[i]); }      // key-derived material: wiped
    delete b;
}

extern "C" int sda_sealedbox_open_rows_dev(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t sk[32], const uint8_t* d_boxes,
                                           size_t slot_bytes, const uint64_t* d_row_bytes, size_t rows, size_t max_box_bytes,
                                           uint8_t* d_out, size_t out_slot, uint64_t* d_out_bytes, uint32_t* d_ok,
                                           uint32_t* d_status, void* stream) {
    if (!b || !pk || !sk) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (rows == 0) return SDA_OK;
    if (!d_boxes || !d_row_bytes || !d_out || !d_out_bytes || !d_status) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (slot_bytes % 16 || out_slot % 16 || !aligned(d_boxes, 16) || !aligned(d_out, 16))
        return capi_fail(SDA_ERR_INVALID_ARGUMENT, "boxes and plaintext rows must be 16-byte aligned (buffers and slots)");
    if (max_box_bytes > slot_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "max_box_bytes exceeds slot_bytes");
    const size_t max_msg = max_box_bytes > SDA_SEALBYTES ? max_box_bytes - SDA_SEALBYTES : 0;
    if (out_slot < max_msg) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "out_slot < max_box_bytes - 48");
    if (hipSetDevice(b->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    if (int st = scratch(b, rows, max_msg)) return st;
    hipError_t e = launch_sealedbox_open(pk, sk, d_boxes, slot_bytes, d_row_bytes, rows, max_box_bytes, d_out, out_slot, d_out_bytes,
                                         d_ok, d_status, static_cast<SboxState*>(b->d_states), static_cast<uint32_t*>(b->d_partial),
                                         reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? SDA_OK : capi_fail(SDA_ERR_HIP, "sealed-box open launch failed: %s", hipGetErrorString(e));
}

extern "C" int sda_sealedbox_seal_rows_dev(sda_sealedbox_t* b, const uint8_t* pks, size_t n_pks, size_t rows_per_key,
                                           const uint8_t* esk, const uint8_t* d_msgs, size_t msg_slot,
                                           const uint64_t* d_msg_bytes, size_t rows, size_t max_msg_bytes, uint8_t* d_boxes,
                                           size_t slot_bytes, uint64_t* d_row_bytes, void* stream) {
    if (!b || !pks || n_pks == 0 || rows_per_key == 0) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "bad recipient key arguments");
    if (rows == 0) return SDA_OK;
    if (!d_msg_bytes || !d_boxes || !d_row_bytes || (max_msg_bytes && !d_msgs)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (slot_bytes % 16 || msg_slot % 16 || !aligned(d_boxes, 16) || !aligned(d_msgs, 16))
        return capi_fail(SDA_ERR_INVALID_ARGUMENT, "messages and boxes must be 16-byte aligned (buffers and slots)");
    if (slot_bytes < max_msg_bytes + SDA_SEALBYTES) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "slot_bytes < max_msg_bytes + 48");
    if (msg_slot < max_msg_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "msg_slot < max_msg_bytes");
    if (hipSetDevice(b->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    if (int st = scratch(b, rows, max_msg_bytes)) return st;
    const size_t key_bytes = rows * 32 + n_pks * 32;
    if (int st = reserve(b->d_keys, b->keys_cap, key_bytes, true)) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    uint8_t* d_esk = static_cast<uint8_t*>(b->d_keys);
    uint8_t* d_pks = d_esk + rows * 32;
    std::vector<uint8_t> fresh;
    if (!esk) {                                             // crypto_box_seal: a fresh key pair per box (OS entropy)
        fresh.resize(rows * 32);
        if (int st = entropy(fresh.data(), fresh.size())) return st;
        esk = fresh.data();
    }
    hipError_t e = hipMemcpyAsync(d_esk, esk, rows * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pks, pks, n_pks * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);       // the host copies of the secrets can go now
    if (!fresh.empty()) explicit_bzero(fresh.data(), fresh.size());
    if (e != hipSuccess) return capi_fail(SDA_ERR_HIP, "uploading the keys failed: %s", hipGetErrorString(e));
    e = launch_sealedbox_seal(d_esk, d_pks, n_pks, rows_per_key, d_msgs, msg_slot, d_msg_bytes, rows, max_msg_bytes, d_boxes,
                              slot_bytes, d_row_bytes, static_cast<SboxState*>(b->d_states), static_cast<uint32_t*>(b->d_partial), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_esk, 0, rows * 32, s);                       // ephemeral secrets are single-use
    return e == hipSuccess ? SDA_OK : capi_fail(SDA_ERR_HIP, "sealed-box seal launch failed: %s", hipGetErrorString(e));
}

// ---- host forms: one payload, staged through the device (what ShareEncryptor::encrypt / ShareDecryptor::decrypt call) ----
extern "C" int sda_sealedbox_seal(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t* esk, const uint8_t* msg, size_t len,
                                  uint8_t* out, size_t out_cap) {
    if (!b || !pk || !out || (len && !msg)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out_cap < len + SDA_SEALBYTES) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small: need %zu bytes", len + SDA_SEALBYTES);
    if (hipSetDevice(b->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    const size_t mslot = (len + 15) / 16 * 16 + 16, bslot = (len + SDA_SEALBYTES + 15) / 16 * 16;
    if (int st = reserve(b->d_io, b->io_cap, mslot + bslot + 32, true)) return st;
    uint8_t* d_msg = static_cast<uint8_t*>(b->d_io);
    uint8_t* d_box = d_msg + mslot;
    uint64_t* d_len = reinterpret_cast<uint64_t*>(d_box + bslot);
    const uint64_t l64 = len;
    if ((len && hipMemcpy(d_msg, msg, len, hipMemcpyHostToDevice) != hipSuccess) || hipMemcpy(d_len, &l64, 8, hipMemcpyHostToDevice) != hipSuccess)
        return capi_fail(SDA_ERR_HIP, "upload failed");
    if (int st = sda_sealedbox_seal_rows_dev(b, pk, 1, 1, esk, d_msg, mslot, d_len, 1, len, d_box, bslot, d_len + 1, nullptr)) return st;
    // the per-row verdict: a recipient key of small order gives the all-zero shared secret, which crypto_box_seal refuses
    // (-1) - the row's length is 0, nothing was encrypted, and the caller gets an error and a wiped buffer, never a box
    uint64_t sealed = 0;
    if (hipMemcpy(&sealed, d_len + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) return capi_fail(SDA_ERR_HIP, "download failed");
    if (sealed != len + SDA_SEALBYTES) {
        (void)hipMemset(d_msg, 0, mslot);
        memset(out, 0, out_cap);
        return capi_fail(SDA_ERR_INVALID_ARGUMENT, "sealing refused: the recipient public key is a small-order point (all-zero shared secret)");
    }
    if (hipMemcpy(out, d_box, len + SDA_SEALBYTES, hipMemcpyDeviceToHost) != hipSuccess) return capi_fail(SDA_ERR_HIP, "download failed");
    (void)hipMemset(d_msg, 0, mslot);
    return SDA_OK;
}

// X25519(sk, base point): the setup pass of a seal writes exactly this as the box's first 32 bytes (its "ephemeral" public key),
// whatever the recipient key is - so an empty message is sealed to the all-zero key and only the epk is read back
extern "C" int sda_sealedbox_public_key(sda_sealedbox_t* b, const uint8_t sk[32], uint8_t pk[32]) {
    if (!b || !sk || !pk) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (hipSetDevice(b->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    if (int st = reserve(b->d_io, b->io_cap, 64 + 32, true)) return st;
    uint8_t* d_box = static_cast<uint8_t*>(b->d_io);
    uint64_t* d_len = reinterpret_cast<uint64_t*>(d_box + 64);
    const uint8_t zero_pk[32] = {0};
    if (hipMemset(d_len, 0, 16) != hipSuccess) return capi_fail(SDA_ERR_HIP, "memset failed");
    if (int st = sda_sealedbox_seal_rows_dev(b, zero_pk, 1, 1, sk, d_box, 16, d_len, 1, 0, d_box, 64, d_len + 1, nullptr)) return st;
    if (hipMemcpy(pk, d_box, 32, hipMemcpyDeviceToHost) != hipSuccess) return capi_fail(SDA_ERR_HIP, "download failed");
    (void)hipMemset(d_box, 0, 64);
    return SDA_OK;
}

extern "C" int sda_sealedbox_open(sda_sealedbox_t* b, const uint8_t pk[32], const uint8_t sk[32], const uint8_t* box, size_t len,
                                  uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!b || !pk || !sk || !out_len || (len && !box)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_len = 0;
    if (len < SDA_SEALBYTES) return capi_fail(SDA_ERR_SODIUM_DECRYPTION, "Sodium decryption failure");          // sodium.rs:80
    const size_t mlen = len - SDA_SEALBYTES;
    if (mlen && (!out || out_cap < mlen)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "output buffer too small: need %zu bytes", mlen);
    if (hipSetDevice(b->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    const size_t bslot = (len + 15) / 16 * 16, mslot = (mlen + 15) / 16 * 16 + 16;
    if (int st = reserve(b->d_io, b->io_cap, mslot + bslot + 32, true)) return st;
    uint8_t* d_box = static_cast<uint8_t*>(b->d_io);
    uint8_t* d_msg = d_box + bslot;
    uint64_t* d_len = reinterpret_cast<uint64_t*>(d_msg + mslot);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_len + 2);
    const uint64_t l64 = len;
    if (hipMemcpy(d_box, box, len, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_len, &l64, 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_flags, 0, 8) != hipSuccess)
        return capi_fail(SDA_ERR_HIP, "upload failed");
    if (int st = sda_sealedbox_open_rows_dev(b, pk, sk, d_box, bslot, d_len, 1, len, d_msg, mslot, d_len + 1, d_flags + 1, d_flags, nullptr)) return st;
    uint32_t flags[2] = {0, 0};
    if (hipMemcpy(flags, d_flags, 8, hipMemcpyDeviceToHost) != hipSuccess) return capi_fail(SDA_ERR_HIP, "download failed");
    if (flags[0] || !flags[1]) { (void)hipMemset(d_msg, 0, mslot); return capi_fail(SDA_ERR_SODIUM_DECRYPTION, "Sodium decryption failure"); }
    if (mlen && hipMemcpy(out, d_msg, mlen, hipMemcpyDeviceToHost) != hipSuccess) return capi_fail(SDA_ERR_HIP, "download failed");
    (void)hipMemset(d_msg, 0, mslot);
    *out_len = mlen;
    return SDA_OK;
}
