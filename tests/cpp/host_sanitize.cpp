// Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (tests/cpp/Makefile
// check-sanitize, run by tests/test_capi_cpu.py without a GPU).  What runs on the host in libsda_hip.so is pointer / size /
// stride arithmetic over caller buffers and, in sda_wire.cpp, a parser of NETWORK input (the SDAJOBv1 clerking-job blob):
// this driver walks every host-only entry point with valid, NULL, short and hostile arguments and mutates job blobs byte by
// byte - any out-of-bounds access, overflow or misaligned load aborts the binary.  Without a device every compute entry
// point must come back with a status (SDA_ERR_NO_DEVICE after its argument checks), never crash.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sda_hip.h"
#include "sda_hip_debug.h"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void schemes() {
    CHECK(sda_scheme_input_size(nullptr) == 0 && sda_scheme_output_size(nullptr) == 0);
    CHECK(sda_scheme_privacy_threshold(nullptr) == 0 && sda_scheme_reconstruction_threshold(nullptr) == 0);
    CHECK(sda_masking_has_mask(nullptr) == 0);
    sda_sharing_scheme_t pss;
    memset(&pss, 0, sizeof pss);
    pss.kind = SDA_SHARING_PACKED_SHAMIR; pss.share_count = 8; pss.modulus = 433; pss.secret_count = 3; pss.privacy_threshold = 4;
    pss.omega_secrets = 354; pss.omega_shares = 150;
    CHECK(sda_scheme_input_size(&pss) == 3 && sda_scheme_output_size(&pss) == 8 && sda_scheme_reconstruction_threshold(&pss) == 7);
    // constructors: argument validation and the host-side matrix algebra run before any device is touched
    sda_share_generator_t* g = nullptr;
    CHECK(sda_share_generator_new(nullptr, &g) != SDA_OK && g == nullptr);
    CHECK(sda_share_generator_new(&pss, nullptr) != SDA_OK);
    int st = sda_share_generator_new(&pss, &g);
    CHECK(st == SDA_OK || st == SDA_ERR_NO_DEVICE);
    if (st == SDA_OK) {
        CHECK(sda_share_generator_share_count(g) == 8 && sda_share_generator_batch_count(g, 10) == 4 && sda_share_generator_rand_count(g, 10) == 16);
        CHECK(sda_share_generator_set_csprng_share_map(g, 7) == SDA_ERR_INVALID_ARGUMENT);
        CHECK(sda_share_generator_generate(g, nullptr, 3, nullptr, 0, nullptr, 8) != SDA_OK);
        int64_t sec[3] = {1, 2, 3}, out[8];
        CHECK(sda_share_generator_generate(g, sec, 3, sec, 3, out, 8) == SDA_ERR_INVALID_ARGUMENT);      // rand_len must be 4
        CHECK(sda_share_generator_generate(g, sec, 3, nullptr, 0, out, 7) == SDA_ERR_INVALID_ARGUMENT);  // out_len must be 8
        sda_share_generator_free(g);
    }
    sda_share_generator_free(nullptr);
    // hostile descriptors (the Aggregation resource travels over the network): every field out of range, wrap-around sums
    const int64_t bad[] = {0, -1, INT64_MIN, INT64_MAX, 1, 2, 4, 4096, 4097, 65535, 65536, (int64_t)1 << 62};
    for (int64_t a : bad)
        for (int64_t b : bad) {
            sda_sharing_scheme_t s = pss;
            s.secret_count = (uint64_t)a; s.privacy_threshold = (uint64_t)b;
            g = nullptr;
            st = sda_share_generator_new(&s, &g);
            CHECK(st != SDA_OK || g != nullptr);
            sda_share_generator_free(g);
            s = pss; s.share_count = (uint64_t)a; s.modulus = b;
            g = nullptr;
            st = sda_share_generator_new(&s, &g);
            sda_share_generator_free(g);
            sda_secret_reconstructor_t* r = nullptr;
            st = sda_secret_reconstructor_new(&s, (size_t)a, &r);
            sda_secret_reconstructor_free(r);
        }
    // tss's shipped parameter sets: the Lagrange matrices (host algebra) for the large shape too
    sda_sharing_scheme_t big = pss;
    big.share_count = 728; big.secret_count = 100; big.privacy_threshold = 155; big.modulus = 746497; big.omega_secrets = 95660; big.omega_shares = 610121;
    g = nullptr;
    st = sda_share_generator_new(&big, &g);
    CHECK(st == SDA_OK || st == SDA_ERR_NO_DEVICE);
    sda_share_generator_free(g);
    sda_masking_scheme_t mk;
    memset(&mk, 0, sizeof mk);
    for (int kind = -1; kind < 5; ++kind)
        for (int64_t m : bad) {
            mk.kind = kind; mk.modulus = m; mk.dimension = 4; mk.seed_bitsize = 128;
            sda_secret_masker_t* sm = nullptr;
            st = sda_secret_masker_new(&mk, &sm);
            sda_secret_masker_free(sm);
            sda_mask_combiner_t* mc = nullptr;
            st = sda_mask_combiner_new(&mk, &mc);
            sda_mask_combiner_free(mc);
        }
}

static void positive_and_sizes() {
    int64_t v[5] = {-1, 0, 432, 433, INT64_MIN}, out[5];
    int st = sda_positive(v, 5, 433, out);
    CHECK(st == SDA_OK || st == SDA_ERR_NO_DEVICE);
    CHECK(sda_positive(nullptr, 5, 433, out) != SDA_OK);
    CHECK(sda_positive(v, 5, 433, out) == SDA_OK && out[0] == 432 && out[1] == 0 && out[3] == 433);   // receive.rs:13-21: no range check
    CHECK(sda_positive(v, 5, -5, out) == SDA_OK && sda_positive(v, 5, INT64_MAX, out) == SDA_OK);      // wraps, never UB
    CHECK(sda_positive(v, 0, 433, nullptr) == SDA_OK);
    CHECK(sda_varint_max_encoded_size(3) == 30 && sda_varint_slot_size(0) == 0);
    (void)sda_varint_slot_size(SIZE_MAX); (void)sda_varint_max_encoded_size(SIZE_MAX);
    (void)sda_base64_encoded_size(SIZE_MAX); (void)sda_base64_decoded_max(SIZE_MAX);
    (void)sda_job_slot_size(SIZE_MAX); (void)sda_job_container_size(SIZE_MAX, 16); (void)sda_job_container_size(SIZE_MAX / 8, SIZE_MAX / 16 * 16);
    CHECK(sda_job_container_size(3, 17) == 0);
    CHECK(sda_strerror(SDA_OK) != nullptr && sda_strerror(-12345) != nullptr && sda_last_error() != nullptr);
    CHECK(sda_debug_set_knob(nullptr, 1) != SDA_OK && sda_debug_set_knob("nope", 1) != SDA_OK);
    char bus[8];
    (void)sda_device_pci_bus_id(0, bus, sizeof bus);            // cap too small: refused, not overrun
    (void)sda_device_pci_bus_id(-1, nullptr, 0);
}

// build a valid job, check it, then mutate: the parser must reject or return a layout whose every row lies inside the blob
static void job_container() {
    const size_t rows = 7, slot = 48;
    const size_t need = sda_job_container_size(rows, slot);
    CHECK(need > 0);
    std::vector<uint8_t> buf(need);
    sda_job_layout_t L;
    CHECK(sda_job_container_init(buf.data(), need - 1, SDA_JOB_VARINT, rows, slot, &L) != SDA_OK);
    CHECK(sda_job_container_init(nullptr, need, SDA_JOB_VARINT, rows, slot, &L) != SDA_OK);
    CHECK(sda_job_container_init(buf.data(), need, SDA_JOB_VARINT, rows, slot, &L) == SDA_OK);
    uint8_t payload[64];
    for (size_t i = 0; i < sizeof payload; ++i) payload[i] = (uint8_t)(i * 3 + 1);
    for (size_t r = 0; r < rows; ++r) CHECK(sda_job_container_set_row(buf.data(), need, r, payload, 5 * r) == SDA_OK);
    CHECK(sda_job_container_set_row(buf.data(), need, rows, payload, 1) != SDA_OK);          // row out of range
    CHECK(sda_job_container_set_row(buf.data(), need, 0, payload, slot + 1) != SDA_OK);      // longer than a slot
    CHECK(sda_job_container_set_row(buf.data(), need - 16, 0, payload, 1) != SDA_OK);        // truncated buffer
    CHECK(sda_job_container_parse(buf.data(), need, &L) == SDA_OK && L.rows == rows && L.slot_bytes == slot);
    for (size_t r = 0; r < rows; ++r) {
        const uint8_t* p = nullptr; size_t len = 0;
        CHECK(sda_job_container_get_row(buf.data(), need, r, &p, &len) == SDA_OK && len == 5 * r && (len == 0 || memcmp(p, payload, len) == 0));
    }
    const uint8_t* p; size_t len;
    CHECK(sda_job_container_get_row(buf.data(), need, rows, &p, &len) != SDA_OK);
    CHECK(sda_job_container_parse(buf.data(), 10, &L) != SDA_OK && sda_job_container_parse(nullptr, need, &L) != SDA_OK);
    CHECK(sda_job_container_parse(buf.data(), need, nullptr) != SDA_OK);
    // truncations: every prefix length
    for (size_t n = 0; n < need; ++n) {
        std::vector<uint8_t> cut(buf.begin(), buf.begin() + n);              // exact-size heap block: ASan sees any overrun
        if (sda_job_container_parse(cut.data(), n, &L) == SDA_OK) CHECK(false);   // a truncated blob never parses
    }
    // mutations of the header and the length table: 20000 blobs with 1-4 bytes / whole u64 fields replaced
    const size_t table_end = 64 + rows * 8;
    for (int trial = 0; trial < 20000; ++trial) {
        std::vector<uint8_t> m(buf);
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits; ++e) {
            const size_t pos = rnd() % table_end;
            if (rnd() & 1) m[pos] = (uint8_t)rnd();
            else {
                const uint64_t vals[] = {0, 1, UINT64_MAX, UINT64_MAX - 15, (uint64_t)1 << 63, need, need + 1, slot, slot + 1, rows + 1, (uint64_t)1 << 40, rnd()};
                const uint64_t v = vals[rnd() % 12];
                memcpy(&m[pos & ~(size_t)7], &v, 8);
            }
        }
        sda_job_layout_t M;
        if (sda_job_container_parse(m.data(), m.size(), &M) == SDA_OK) {
            // whatever parsed must describe memory inside the blob
            CHECK(M.payload_offset <= m.size() && M.lengths_offset <= m.size());
            CHECK(M.rows == 0 || (M.slot_bytes % 16 == 0 && M.rows <= (m.size() - M.payload_offset) / (M.slot_bytes ? M.slot_bytes : 1)));
            for (size_t r = 0; r < M.rows && r < 64; ++r) {
                const uint8_t* q = nullptr; size_t l = 0;
                if (sda_job_container_get_row(m.data(), m.size(), r, &q, &l) == SDA_OK && l) {
                    CHECK(q >= m.data() && q + l <= m.data() + m.size());
                    volatile uint8_t sink = q[0] ^ q[l - 1];                 // touch both ends under ASan
                    (void)sink;
                }
            }
        }
    }
}

static void handles_without_device_or_with_null() {
    // every trait-shaped host call with NULL handles / buffers: a status, never a crash
    int64_t x[4] = {1, 2, 3, 4}, out[8];
    size_t n = 0;
    const int64_t* rows[2] = {x, x};
    size_t lens[2] = {4, 3};
    CHECK(sda_share_combiner_combine(nullptr, rows, lens, 2, out, 8, &n) != SDA_OK);
    CHECK(sda_share_generator_generate(nullptr, x, 4, nullptr, 0, out, 8) != SDA_OK);
    CHECK(sda_secret_reconstructor_reconstruct(nullptr, nullptr, nullptr, nullptr, 0, out, 4, &n) != SDA_OK);
    CHECK(sda_secret_masker_mask(nullptr, x, 4, nullptr, 0, out, 4, &n, out) != SDA_OK);
    CHECK(sda_mask_combiner_combine(nullptr, rows, lens, 2, out, 8, &n) != SDA_OK);
    CHECK(sda_secret_unmasker_unmask(nullptr, x, 4, x, 4, out) != SDA_OK);
    CHECK(sda_share_combiner_begin(nullptr, 4) != SDA_OK && sda_share_combiner_update(nullptr, x, 1, 4) != SDA_OK && sda_share_combiner_finish(nullptr, out) != SDA_OK);
    sda_sharing_scheme_t add;
    memset(&add, 0, sizeof add);
    add.kind = SDA_SHARING_ADDITIVE; add.share_count = 3; add.modulus = 433;
    sda_share_combiner_t* c = nullptr;
    int st = sda_share_combiner_new(&add, &c);
    CHECK(st == SDA_OK || st == SDA_ERR_NO_DEVICE);
    if (st == SDA_OK) {
        CHECK(sda_share_combiner_combine(c, rows, lens, 2, out, 8, &n) == SDA_ERR_WRONG_DIMENSION);       // combiner.rs:21
        CHECK(sda_share_combiner_combine(c, rows, lens, 0, out, 8, &n) == SDA_OK && n == 0);              // combiner.rs:17
        lens[1] = 4;
        CHECK(sda_share_combiner_combine(c, rows, lens, 2, out, 3, &n) != SDA_OK);                         // output too small
        CHECK(sda_share_combiner_update(c, x, 1, 4) == SDA_ERR_STATE);
        CHECK(sda_share_combiner_set_residency(c, 9) != SDA_OK);
        sda_share_combiner_free(c);
    }
}

int main() {
    schemes();
    positive_and_sizes();
    job_container();
    handles_without_device_or_with_null();
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("host side clean under ASan + UBSan (%s)\n", sda_version());
    return 0;
}
