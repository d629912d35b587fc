// Binary clerking-job container and base64 of `Binary` payloads (include/sda_hip.h "Clerking-job container";
// SURVEY.md 8f rank 3).  The container functions are host-only byte layout code (no device needed); the base64 calls
// launch wire_kernels.hip.
#include "../../include/sda_hip.h"

#include <hip/hip_runtime_api.h>
#include <string.h>

#include "capi_internal.hpp"
#include "kernels.hpp"

using namespace sda;

namespace {
const char kMagic[8] = {'S', 'D', 'A', 'J', 'O', 'B', 'v', '1'};
const size_t kHeader = 64;

void put32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
void put64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
uint32_t get32(const uint8_t* p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); return v; }
uint64_t get64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

// rows * slot + table + header without overflow
bool container_geometry(uint64_t rows, uint64_t slot, uint64_t& lengths_off, uint64_t& payload_off, uint64_t& total) {
    if (slot % 16 || rows > (1ull << 40) || slot > (1ull << 40)) return false;
    lengths_off = kHeader;
    payload_off = (kHeader + rows * 8 + 15) / 16 * 16;
    if (slot && rows > (UINT64_MAX - payload_off) / slot) return false;
    total = payload_off + rows * slot;
    return true;
}
}  // namespace

extern "C" size_t sda_job_slot_size(size_t max_payload_bytes) { return (max_payload_bytes + 15) / 16 * 16; }

extern "C" size_t sda_job_container_size(size_t rows, size_t slot_bytes) {
    uint64_t lo, po, total;
    return container_geometry(rows, slot_bytes, lo, po, total) ? (size_t)total : 0;
}

extern "C" int sda_job_container_init(uint8_t* buf, size_t cap, uint32_t payload_kind, size_t rows, size_t slot_bytes,
                                      sda_job_layout_t* out) {
    if (!buf) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "buf is NULL");
    if (payload_kind > SDA_JOB_BASE64_TEXT) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "unknown payload kind %u", payload_kind);
    uint64_t lo, po, total;
    if (!container_geometry(rows, slot_bytes, lo, po, total))
        return capi_fail(SDA_ERR_INVALID_ARGUMENT, "slot_bytes must be a multiple of 16 (sda_job_slot_size) and the job must fit 2^64 bytes");
    if (cap < total) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "buffer too small: the container needs %llu bytes", (unsigned long long)total);
    memset(buf, 0, (size_t)po);                       // header + length table (+ alignment gap)
    memcpy(buf, kMagic, 8);
    put32(buf + 8, (uint32_t)kHeader);
    put32(buf + 12, payload_kind);
    put64(buf + 16, rows);
    put64(buf + 24, slot_bytes);
    put64(buf + 32, lo);
    put64(buf + 40, po);
    put64(buf + 48, total);
    put64(buf + 56, 0);
    if (out) { out->payload_kind = payload_kind; out->rows = rows; out->slot_bytes = slot_bytes; out->lengths_offset = lo; out->payload_offset = po; out->total_bytes = total; }
    return SDA_OK;
}

namespace {
// header + geometry + total size of an untrusted blob: everything but the length table (O(1))
int parse_header(const uint8_t* buf, size_t n_bytes, sda_job_layout_t* out) {
    if (!buf || !out) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_bytes < kHeader || memcmp(buf, kMagic, 8) != 0) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "not an SDAJOBv1 container");
    if (get32(buf + 8) != kHeader) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "unsupported header size %u", get32(buf + 8));
    const uint32_t kind = get32(buf + 12);
    const uint64_t rows = get64(buf + 16), slot = get64(buf + 24);
    uint64_t lo, po, total;
    if (kind > SDA_JOB_BASE64_TEXT || !container_geometry(rows, slot, lo, po, total) || get64(buf + 32) != lo || get64(buf + 40) != po ||
        get64(buf + 48) != total || get64(buf + 56) != 0)
        return capi_fail(SDA_ERR_INVALID_ARGUMENT, "inconsistent SDAJOBv1 header");
    if (n_bytes < total) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "truncated container: header says %llu bytes, have %zu", (unsigned long long)total, n_bytes);
    out->payload_kind = kind; out->rows = rows; out->slot_bytes = slot; out->lengths_offset = lo; out->payload_offset = po; out->total_bytes = total;
    return SDA_OK;
}
}  // namespace

extern "C" int sda_job_container_parse(const uint8_t* buf, size_t n_bytes, sda_job_layout_t* out) {
    sda_job_layout_t L;
    if (int st = parse_header(buf, n_bytes, out ? &L : nullptr)) return st;
    for (uint64_t r = 0; r < L.rows; ++r)
        if (get64(buf + L.lengths_offset + 8 * r) > L.slot_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "row %llu is longer than its slot", (unsigned long long)r);
    *out = L;
    return SDA_OK;
}

// the buffer may be a parsed, untrusted blob: its header is re-validated against `cap` before a byte is written
extern "C" int sda_job_container_set_row(uint8_t* buf, size_t cap, size_t row, const uint8_t* payload, size_t len) {
    sda_job_layout_t L;
    if (int st = parse_header(buf, cap, &L)) return st;
    if (row >= L.rows) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "row %zu out of range (%llu rows)", row, (unsigned long long)L.rows);
    if (len > L.slot_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "payload of %zu bytes does not fit the %llu-byte slot", len, (unsigned long long)L.slot_bytes);
    if (len && !payload) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "payload is NULL");
    uint8_t* dst = buf + L.payload_offset + row * L.slot_bytes;
    if (len) memcpy(dst, payload, len);
    memset(dst + len, 0, (size_t)(L.slot_bytes - len));        // the tail of a slot is defined (zero), never stale bytes
    put64(buf + L.lengths_offset + 8 * row, len);
    return SDA_OK;
}

// O(1) per row: the header is validated (geometry, total size) and this row's length against its slot - iterating a
// 100k-row job through this call does not rescan the length table every time
extern "C" int sda_job_container_get_row(const uint8_t* buf, size_t n_bytes, size_t row, const uint8_t** payload, size_t* len) {
    sda_job_layout_t L;
    if (int st = parse_header(buf, n_bytes, &L)) return st;
    if (!payload || !len) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (row >= L.rows) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "row %zu out of range", row);
    const uint64_t n = get64(buf + L.lengths_offset + 8 * row);
    if (n > L.slot_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "row %zu is longer than its slot", row);
    *payload = buf + L.payload_offset + row * L.slot_bytes;
    *len = (size_t)n;
    return SDA_OK;
}

// ---- base64 ------------------------------------------------------------------------------------------------
extern "C" size_t sda_base64_encoded_size(size_t n_bytes) { return (n_bytes + 2) / 3 * 4; }
extern "C" size_t sda_base64_decoded_max(size_t n_chars) { return n_chars / 4 * 3; }

extern "C" int sda_base64_decode_rows_dev(const uint8_t* d_text, const uint64_t* d_text_offsets, size_t text_slot,
                                          const uint64_t* d_text_bytes, size_t rows, size_t max_chars, uint8_t* d_out,
                                          size_t out_slot, uint64_t* d_out_bytes, uint32_t* d_status, uint32_t* d_row_status,
                                          void* stream) {
    if (int st = capi_device_ready()) return st;
    if (rows == 0) return SDA_OK;
    if (!d_text || !d_text_bytes || !d_out || !d_out_bytes || !d_status) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (out_slot % 4 || ((uintptr_t)d_out & 3u)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "output rows must be 4-byte aligned (buffer and out_slot)");
    if (out_slot < sda_base64_decoded_max(max_chars)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "out_slot < 3/4 of max_chars");
    if (!d_text_offsets && text_slot < max_chars) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "text_slot < max_chars");
    hipError_t e = launch_base64_decode_rows(d_text, d_text_offsets, text_slot, d_text_bytes, rows, max_chars, d_out, out_slot, d_out_bytes,
                                             d_status, d_row_status, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? SDA_OK : capi_fail(SDA_ERR_HIP, "base64 decode launch failed: %s", hipGetErrorString(e));
}

extern "C" int sda_base64_encode_rows_dev(const uint8_t* d_in, size_t in_slot, const uint64_t* d_in_bytes, size_t rows,
                                          size_t max_bytes, uint8_t* d_text, size_t text_slot, uint64_t* d_text_bytes,
                                          void* stream) {
    if (int st = capi_device_ready()) return st;
    if (rows == 0) return SDA_OK;
    if (!d_in || !d_in_bytes || !d_text || !d_text_bytes) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (in_slot % 4 || ((uintptr_t)d_in & 3u)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "input rows must be 4-byte aligned (buffer and in_slot)");
    if (text_slot % 16 || ((uintptr_t)d_text & 15u)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "text rows must be 16-byte aligned (buffer and text_slot)");
    if (text_slot < sda_base64_encoded_size(max_bytes)) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "text_slot < sda_base64_encoded_size(max_bytes)");
    hipError_t e = launch_base64_encode_rows(d_in, in_slot, d_in_bytes, rows, max_bytes, d_text, text_slot, d_text_bytes,
                                             reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? SDA_OK : capi_fail(SDA_ERR_HIP, "base64 encode launch failed: %s", hipGetErrorString(e));
}
