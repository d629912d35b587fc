/* The reference's README walkthrough (README.md:86-157: dimension 10, modulus 433, additive sharing over 3 clerks,
 * three participants submitting 0..9, zeros and 0 1 0 1 ..., expected result 0 2 2 4 4 6 6 8 8 10) and its
 * packed-Shamir integration vector (full_loop.rs:54-67,148) through nothing but the C ABI of include/sda_hip.h -
 * the calls a Rust `extern "C"` shim, cgo or any other FFI would make - and a clerk's job with sealed payloads
 * (varint + sealed box per participant, one SDAJOBv1 blob, open -> streaming clerk sum).  Plain C99.
 *
 *   gcc -std=c99 -I include examples/c_abi_walkthrough.c -L sda_amd/lib -lsda_hip -Wl,-rpath,$PWD/sda_amd/lib -o walkthrough
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sda_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int st_ = (call);                                                                        \
        if (st_ != SDA_OK) {                                                                     \
            fprintf(stderr, "%s -> %s: %s\n", #call, sda_strerror(st_), sda_last_error());       \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static int run(const sda_sharing_scheme_t* scheme, const sda_masking_scheme_t* masking, size_t dim,
               const int64_t* inputs, size_t participants, const size_t* clerks, size_t n_clerks_used, int64_t* result) {
    sda_share_generator_t* gen;
    sda_share_combiner_t* comb;
    sda_secret_reconstructor_t* rec;
    sda_secret_masker_t* masker;
    sda_mask_combiner_t* mask_comb;
    sda_secret_unmasker_t* unmasker;
    CHECK(sda_share_generator_new(scheme, &gen));
    CHECK(sda_share_combiner_new(scheme, &comb));
    CHECK(sda_secret_reconstructor_new(scheme, dim, &rec));
    CHECK(sda_secret_masker_new(masking, &masker));
    CHECK(sda_mask_combiner_new(masking, &mask_comb));
    CHECK(sda_secret_unmasker_new(masking, &unmasker));
    const size_t n = (size_t)sda_share_generator_share_count(gen), B = (size_t)sda_share_generator_batch_count(gen, dim);
    const size_t mask_len = (size_t)sda_secret_masker_mask_len(masker, dim);

    /* participants (participate.rs:52-76): mask, then share the masked secrets; shares[p][clerk][B] */
    int64_t* shares = malloc(participants * n * B * sizeof(int64_t));
    int64_t* masks = malloc(participants * (mask_len ? mask_len : 1) * sizeof(int64_t));
    int64_t* masked = malloc(dim * sizeof(int64_t));
    for (size_t p = 0; p < participants; ++p) {
        size_t got = 0;
        CHECK(sda_secret_masker_mask(masker, inputs + p * dim, dim, NULL, 0, masks + p * mask_len, mask_len, &got, masked));
        CHECK(sda_share_generator_generate(gen, masked, dim, NULL, 0, shares + p * n * B, n * B));
    }
    /* clerks (clerk.rs:85-86): each sums its column of every participant's shares */
    int64_t* sums = malloc(n * B * sizeof(int64_t));
    const int64_t** rows = malloc(participants * sizeof(*rows));
    size_t* lens = malloc(participants * sizeof(*lens));
    for (size_t c = 0; c < n; ++c) {
        size_t out_len = 0;
        for (size_t p = 0; p < participants; ++p) { rows[p] = shares + (p * n + c) * B; lens[p] = B; }
        CHECK(sda_share_combiner_combine(comb, rows, lens, participants, sums + c * B, B, &out_len));
    }
    /* recipient (receive.rs:113-152): combine masks, reconstruct from the clerks that answered, unmask, positive */
    int64_t* mask_total = malloc((dim ? dim : 1) * sizeof(int64_t));
    size_t mask_total_len = 0;
    for (size_t p = 0; p < participants; ++p) { rows[p] = masks + p * mask_len; lens[p] = mask_len; }
    CHECK(sda_mask_combiner_combine(mask_comb, rows, lens, participants, mask_total, dim, &mask_total_len));
    const int64_t** srows = malloc(n_clerks_used * sizeof(*srows));
    size_t* slens = malloc(n_clerks_used * sizeof(*slens));
    for (size_t i = 0; i < n_clerks_used; ++i) { srows[i] = sums + clerks[i] * B; slens[i] = B; }
    int64_t* masked_total = malloc(dim * sizeof(int64_t));
    size_t n_out = 0;
    CHECK(sda_secret_reconstructor_reconstruct(rec, clerks, srows, slens, n_clerks_used, masked_total, dim, &n_out));
    int64_t* unmasked = malloc(dim * sizeof(int64_t));
    CHECK(sda_secret_unmasker_unmask(unmasker, mask_total, mask_total_len, masked_total, n_out, unmasked));
    CHECK(sda_positive(unmasked, dim, scheme->modulus, result));

    free(shares); free(masks); free(masked); free(sums); free(rows); free(lens); free(mask_total); free(srows);
    free(slens); free(masked_total); free(unmasked);
    sda_share_generator_free(gen); sda_share_combiner_free(comb); sda_secret_reconstructor_free(rec);
    sda_secret_masker_free(masker); sda_mask_combiner_free(mask_comb); sda_secret_unmasker_free(unmasker);
    return 0;
}

int main(void) {
    if (sda_device_count() < 1) { printf("no GPU: %s\n", sda_strerror(SDA_ERR_NO_DEVICE)); return 2; }
    int failures = 0;
    int64_t out[10];
    {   /* README walkthrough */
        sda_sharing_scheme_t additive;
        memset(&additive, 0, sizeof additive);
        additive.kind = SDA_SHARING_ADDITIVE; additive.share_count = 3; additive.modulus = 433;
        sda_masking_scheme_t none;
        memset(&none, 0, sizeof none);
        none.kind = SDA_MASKING_NONE;
        const int64_t inputs[3][10] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, {0}, {0, 1, 0, 1, 0, 1, 0, 1, 0, 1}};
        const size_t all[3] = {0, 1, 2};
        const int64_t want[10] = {0, 2, 2, 4, 4, 6, 6, 8, 8, 10};
        if (run(&additive, &none, 10, &inputs[0][0], 3, all, 3, out)) return 1;
        if (memcmp(out, want, sizeof want)) { printf("FAIL: README walkthrough\n"); ++failures; }
    }
    {   /* full_loop.rs with_packedshamir + a ChaCha mask, one clerk missing */
        sda_sharing_scheme_t pss;
        memset(&pss, 0, sizeof pss);
        pss.kind = SDA_SHARING_PACKED_SHAMIR; pss.share_count = 8; pss.modulus = 433; pss.secret_count = 3;
        pss.privacy_threshold = 4; pss.omega_secrets = 354; pss.omega_shares = 150;
        sda_masking_scheme_t chacha;
        memset(&chacha, 0, sizeof chacha);
        chacha.kind = SDA_MASKING_CHACHA; chacha.modulus = 433; chacha.dimension = 4; chacha.seed_bitsize = 128;
        const int64_t inputs[2][4] = {{1, 2, 3, 4}, {1, 2, 3, 4}};
        const size_t answered[7] = {7, 5, 4, 3, 2, 1, 0};
        const int64_t want[4] = {2, 4, 6, 8};
        if (run(&pss, &chacha, 4, &inputs[0][0], 2, answered, 7, out)) return 1;
        if (memcmp(out, want, sizeof want)) { printf("FAIL: packed Shamir + ChaCha mask\n"); ++failures; }
    }
    {   /* a clerk's job with the payloads sealed (participate.rs:82-101 -> stores.rs:86-101 -> clerk.rs:78-86): every
         * participant varint-encodes and seals its share vector for clerk 0, the "server" packs the encryptions into one
         * SDAJOBv1 blob, the clerk parses it, opens every box and streams the payloads into its running sum */
        sda_sharing_scheme_t additive;
        memset(&additive, 0, sizeof additive);
        additive.kind = SDA_SHARING_ADDITIVE; additive.share_count = 3; additive.modulus = 433;
        sda_share_generator_t* gen; sda_share_combiner_t* comb; sda_varint_codec_t* codec; sda_sealedbox_t* sbox;
        CHECK(sda_share_generator_new(&additive, &gen));
        CHECK(sda_share_combiner_new(&additive, &comb));
        CHECK(sda_varint_codec_new(&codec));
        CHECK(sda_sealedbox_new(&sbox));
        uint8_t sk[32], pk[32];
        for (int i = 0; i < 32; ++i) sk[i] = (uint8_t)(3 * i + 7);
        /* the public key X25519(sk, 9) (key generation itself stays with the reference's keystore) */
        CHECK(sda_sealedbox_public_key(sbox, sk, pk));
        const int64_t inputs[3][10] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, {0}, {0, 1, 0, 1, 0, 1, 0, 1, 0, 1}};
        int64_t shares[3][10], clear_sum[10];
        uint8_t wire[128], boxes[3][128 + SDA_SEALBYTES];
        size_t box_len[3];
        const size_t slot = sda_job_slot_size(128 + SDA_SEALBYTES);
        uint8_t* job = malloc(sda_job_container_size(3, slot));
        sda_job_layout_t layout;
        CHECK(sda_job_container_init(job, sda_job_container_size(3, slot), SDA_JOB_SEALED, 3, slot, &layout));
        for (size_t p = 0; p < 3; ++p) {
            int64_t all_shares[3 * 10];
            size_t n_wire = 0;
            CHECK(sda_share_generator_generate(gen, inputs[p], 10, NULL, 0, all_shares, 30));
            memcpy(shares[p], all_shares, sizeof shares[p]);                                  /* clerk 0's vector */
            CHECK(sda_varint_encode(codec, shares[p], 10, wire, sizeof wire, &n_wire));       /* sodium.rs:36-41 */
            CHECK(sda_sealedbox_seal(sbox, pk, NULL, wire, n_wire, boxes[p], sizeof boxes[p]));   /* sodium.rs:43 */
            box_len[p] = n_wire + SDA_SEALBYTES;
            CHECK(sda_job_container_set_row(job, layout.total_bytes, p, boxes[p], box_len[p]));
        }
        sda_job_layout_t got;
        CHECK(sda_job_container_parse(job, layout.total_bytes, &got));
        CHECK(sda_share_combiner_begin(comb, 10));
        for (size_t p = 0; p < got.rows; ++p) {
            const uint8_t* enc; size_t enc_len = 0, n_plain = 0;
            uint8_t plain[128];
            CHECK(sda_job_container_get_row(job, layout.total_bytes, p, &enc, &enc_len));
            CHECK(sda_sealedbox_open(sbox, pk, sk, enc, enc_len, plain, sizeof plain, &n_plain));   /* sodium.rs:78 */
            CHECK(sda_share_combiner_update_varint(comb, codec, plain, n_plain));              /* decode -> add -> discard */
        }
        int64_t sum[10];
        CHECK(sda_share_combiner_finish(comb, sum));
        {   const int64_t* rows[3] = {shares[0], shares[1], shares[2]};
            const size_t lens[3] = {10, 10, 10};
            size_t n_out = 0;
            CHECK(sda_share_combiner_combine(comb, rows, lens, 3, clear_sum, 10, &n_out)); }
        if (memcmp(sum, clear_sum, sizeof sum)) { printf("FAIL: sealed clerking job\n"); ++failures; }
        boxes[1][40] ^= 1;                                                                     /* a tampered encryption */
        {   uint8_t plain[128]; size_t n_plain = 0;
            if (sda_sealedbox_open(sbox, pk, sk, boxes[1], box_len[1], plain, sizeof plain, &n_plain) != SDA_ERR_SODIUM_DECRYPTION ||
                strcmp(sda_last_error(), "Sodium decryption failure") != 0) { printf("FAIL: tamper not detected\n"); ++failures; } }
        free(job);
        sda_share_generator_free(gen); sda_share_combiner_free(comb); sda_varint_codec_free(codec); sda_sealedbox_free(sbox);
    }
    if (!failures) printf("c_abi_walkthrough: OK\n");
    return failures ? 1 : 0;
}
