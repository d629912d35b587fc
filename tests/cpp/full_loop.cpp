// C++ rendition of the reference's integration-tests/tests/full_loop.rs (:29-67, :113, :148) and of
// the README walkthrough (README.md:86,105-107,157), driving the HIP core through the C++ host mirror
// with OS randomness exactly like the reference tests do (no injection): participants -> snapshot
// transposition -> clerks -> reveal, asserting the revealed output.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "sda_crypto.hpp"

using namespace sda_client;

struct Aggregation {                       // protocol/src/resources.rs:44-67 (compute-relevant fields)
    size_t vector_dimension;
    int64_t modulus;
    LinearMaskingScheme masking_scheme;
    LinearSecretSharingScheme committee_sharing_scheme;
};

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL %s: %s\n", what, #cond); ++failures; } } while (0)

static std::vector<int64_t> run(const Aggregation& a, const std::vector<std::vector<int64_t>>& inputs,
                                const std::vector<size_t>* clerk_subset = nullptr) {
    CryptoModule crypto;
    const size_t n = a.committee_sharing_scheme.output_size();
    std::vector<std::vector<Mask>> masks;
    std::vector<std::vector<std::vector<Share>>> per_participant;           // [p][clerk][batch]
    for (const auto& secrets : inputs) {                                    // participate.rs:52-76
        auto masker = crypto.new_secret_masker(a.masking_scheme);
        auto mm = masker->mask(secrets);
        if (a.masking_scheme.has_mask()) masks.push_back(mm.first);
        auto generator = crypto.new_share_generator(a.committee_sharing_scheme);
        per_participant.push_back(generator->generate(mm.second));
    }
    std::vector<std::vector<Share>> clerk_sums(n);
    for (size_t c = 0; c < n; ++c) {                                        // stores.rs:86-101 + clerk.rs:85-86
        std::vector<std::vector<Share>> job;
        for (auto& pp : per_participant) job.push_back(pp[c]);
        clerk_sums[c] = crypto.new_share_combiner(a.committee_sharing_scheme)->combine(job);
    }
    std::vector<Mask> mask;                                                 // receive.rs:102-118
    if (a.masking_scheme.has_mask()) mask = crypto.new_mask_combiner(a.masking_scheme)->combine(masks);
    std::vector<std::pair<size_t, std::vector<Share>>> indexed;             // receive.rs:127-138
    if (clerk_subset) for (size_t c : *clerk_subset) indexed.push_back({c, clerk_sums[c]});
    else for (size_t c = 0; c < n; ++c) indexed.push_back({c, clerk_sums[c]});
    auto masked_output = crypto.new_secret_reconstructor(a.committee_sharing_scheme, a.vector_dimension)->reconstruct(indexed);
    auto output = crypto.new_secret_unmasker(a.masking_scheme)->unmask({mask, masked_output});   // receive.rs:149-152
    return RecipientOutput{a.modulus, output}.positive().values;
}

int main() {
    if (sda_device_count() < 1) { printf("no GPU: %s\n", sda_strerror(SDA_ERR_NO_DEVICE)); return 2; }
    const auto add = LinearSecretSharingScheme::Additive(3, 433);
    const auto pss = LinearSecretSharingScheme::PackedShamir(3, 8, 4, 433, 354, 150);
    const std::vector<std::vector<int64_t>> two = {{1, 2, 3, 4}, {1, 2, 3, 4}};
    const std::vector<int64_t> want = {2, 4, 6, 8};
    EXPECT(run({4, 433, LinearMaskingScheme::None(), add}, two) == want, "simple (full_loop.rs:29-32)");
    EXPECT(run({4, 433, LinearMaskingScheme::Full(433), add}, two) == want, "with_fullmask (:34-40)");
    EXPECT(run({4, 433, LinearMaskingScheme::ChaCha(433, 4, 128), add}, two) == want, "with_chachamask (:42-52)");
    EXPECT(run({4, 433, LinearMaskingScheme::None(), pss}, two) == want, "with_packedshamir (:54-67)");
    const std::vector<size_t> subset = {7, 5, 4, 3, 2, 1, 0};
    EXPECT(run({4, 433, LinearMaskingScheme::Full(433), pss}, two, &subset) == want, "packed shamir, one clerk missing");
    // README walkthrough: dim 10, q 433, 3 shares, 3 participants
    const std::vector<std::vector<int64_t>> readme = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 1, 0, 1, 0, 1, 0, 1, 0, 1}};
    EXPECT(run({10, 433, LinearMaskingScheme::None(), add}, readme) == (std::vector<int64_t>{0, 2, 2, 4, 4, 6, 6, 8, 8, 10}),
           "README walkthrough (README.md:157)");
    // the reference's own value representation (Rust's truncated %): additive.rs:42-47 leaves the n-1 draws as they are
    // and folds the last share into (-q, q) - here 1 - r1 - r2 is negative unless both draws are tiny - and the clerks'
    // and the recipient's sums keep the sign of their running totals; positive() lands on the same output
    {
        ShareGenerator gen(add);
        gen.set_value_mode(SDA_VALUES_RUST_SIGNED);
        const std::vector<int64_t> secrets = {1, 2, 3, 4};
        const auto shares = gen.generate(secrets);
        bool shape = shares.size() == 3, fold = true, some_negative = false;
        for (size_t i = 0; shape && i < 4; ++i) {
            fold = fold && shares[0][i] >= 0 && shares[0][i] < 433 && shares[1][i] >= 0 && shares[1][i] < 433 &&
                   shares[2][i] == ((secrets[i] - shares[0][i]) % 433 - shares[1][i]) % 433;       // C's % truncates like Rust's
            some_negative = some_negative || shares[2][i] < 0;
        }
        EXPECT(shape && fold, "rust_signed additive shares are the reference's fold");
        EXPECT(some_negative, "rust_signed: the folded share keeps Rust's sign");
        ShareCombiner comb(add);
        comb.set_value_mode(SDA_VALUES_RUST_SIGNED);
        std::vector<std::pair<size_t, std::vector<Share>>> sums;
        for (size_t c = 0; c < 3; ++c) sums.push_back({c, comb.combine({shares[c], shares[c]})});     // two identical participants
        SecretReconstructor rec(add, 4);
        rec.set_value_mode(SDA_VALUES_RUST_SIGNED);
        EXPECT((RecipientOutput{433, rec.reconstruct(sums)}.positive().values) == want, "rust_signed round trip");
    }
    // the same loop over the wire format: every share vector varint-encoded (sodium.rs:36-41), each clerk streaming
    // the opened payloads into its running sum (clerk.rs:78-86 without materialising them)
    {
        ShareCodec codec;
        ShareGenerator gen(pss);
        std::vector<std::vector<std::vector<uint8_t>>> wire(8);           // [clerk][participant] payload
        for (const auto& secrets : two) {
            const auto shares = gen.generate(secrets);
            for (size_t c = 0; c < 8; ++c) {
                wire[c].push_back(codec.encode(shares[c]));
                EXPECT(codec.decode(wire[c].back()) == shares[c], "codec round trip");
            }
        }
        std::vector<std::pair<size_t, std::vector<int64_t>>> sums;
        StreamingShareCombiner clerk(pss);
        for (size_t c = 0; c < 8; ++c) {
            clerk.begin(wire[c][0].empty() ? 0 : codec.decode(wire[c][0]).size());
            for (const auto& payload : wire[c]) clerk.add(codec, payload);
            sums.push_back({c, clerk.finish()});
        }
        auto out = SecretReconstructor(pss, 4).reconstruct(sums);
        EXPECT((RecipientOutput{433, out}.positive().values) == want, "packed shamir over the wire format, streaming clerks");
        try {
            clerk.begin(2);
            clerk.add(codec, codec.encode({1, 2, 3}));
            EXPECT(false, "a payload of the wrong length must fail");
        } catch (const SdaClientError& e) { EXPECT(std::string(e.what()) == "Wrong dimension", "combiner.rs:21 on the wire form"); }
    }
    // the two parametrisations of a sharing drawn by the library itself (sda_hip.h "CSPRNG share map"): both reveal the same sums
    {
        for (int map : {SDA_SHARE_MAP_SYSTEMATIC, SDA_SHARE_MAP_TSS_NODES}) {
            ShareGenerator gen(pss);
            EXPECT(gen.csprng_share_map() == SDA_SHARE_MAP_SYSTEMATIC, "matrix-form kernels default to the systematic map");
            gen.set_csprng_share_map(map);
            EXPECT(gen.csprng_share_map() == map, "share map setter");
            ShareCombiner comb(pss);
            std::vector<std::vector<std::vector<Share>>> per;            // [participant][clerk]
            for (const auto& secrets : two) per.push_back(gen.generate(secrets));
            std::vector<std::pair<size_t, std::vector<int64_t>>> sums;
            for (size_t c : {7u, 1u, 2u, 3u, 4u, 5u, 0u}) sums.push_back({c, comb.combine({per[0][c], per[1][c]})});
            auto out = SecretReconstructor(pss, 4).reconstruct(sums);
            EXPECT((RecipientOutput{433, out}.positive().values) == want, "reveal is independent of the share map");
        }
    }
    // the reference's full path for a clerk: participants seal their varint-encoded share vectors to the clerk's key
    // (participate.rs:82-101, sodium.rs:33-46), the server hands the clerk ONE job (here an SDAJOBv1 blob instead of a
    // JSON array, stores.rs:86-101), the clerk decrypts and combines (clerk.rs:78-86, sodium.rs:72-92)
    {
        // a key pair: the public key is X25519(sk, 9)
        DecryptionKey sk(32);
        for (int i = 0; i < 32; ++i) sk[i] = (uint8_t)(7 * i + 1);
        SealedBox kb;
        const EncryptionKey pk = kb.public_key(sk);
        // sealing to a small-order key (all-zero shared secret) is refused like crypto_box_seal's -1, never a box anyone can open
        bool refused = false;
        try { kb.seal({1, 2, 3}, EncryptionKey(32, 0)); } catch (const SdaClientError&) { refused = true; }
        EXPECT(refused, "seal to the all-zero public key is refused");
        ShareGenerator gen(add);
        ShareEncryptor enc(pk);
        ShareDecryptor dec(pk, sk);
        std::vector<std::vector<uint8_t>> job_rows;
        std::vector<std::vector<int64_t>> clear;
        for (const auto& secrets : two) {
            const auto shares = gen.generate(secrets);
            clear.push_back(shares[0]);
            job_rows.push_back(enc.encrypt(shares[0]));                 // clerk 0's encryption of this participation
            EXPECT(job_rows.back().size() > SDA_SEALBYTES, "sealed payload");
        }
        const JobContainer sent = JobContainer::build(SDA_JOB_SEALED, job_rows);
        const JobContainer got = JobContainer::parse(sent.blob);
        EXPECT(got.layout.rows == 2 && got.layout.payload_kind == SDA_JOB_SEALED, "job container header");
        std::vector<std::vector<int64_t>> opened;
        for (size_t r = 0; r < got.layout.rows; ++r) opened.push_back(dec.decrypt(got.row(r)));
        EXPECT(opened == clear, "seal -> SDAJOBv1 -> open -> decode round trip");
        EXPECT(ShareCombiner(add).combine(opened) == ShareCombiner(add).combine(clear), "clerk sums over decrypted shares");
        Encryption bad = job_rows[0];
        bad[40] ^= 1;
        try {
            dec.decrypt(bad);
            EXPECT(false, "a tampered encryption must fail");
        } catch (const SdaClientError& e) { EXPECT(std::string(e.what()) == "Sodium decryption failure", "sodium.rs:80"); }
    }
    // error behaviour mirrors the reference's strings
    try {
        ShareCombiner(add).combine({{1, 2, 3}, {1, 2}});
        EXPECT(false, "ragged combine must fail");
    } catch (const SdaClientError& e) { EXPECT(std::string(e.what()) == "Wrong dimension", "combiner.rs:21"); }
    try {
        SecretReconstructor(pss, 4).reconstruct({{0, {1, 2}}, {1, {1, 2}}});
        EXPECT(false, "too few shares must fail");
    } catch (const SdaClientError& e) { EXPECT(std::string(e.what()) == "Not enough shares to reconstruct", "packed_shamir.rs:75"); }
    try {
        SecretMasker(LinearMaskingScheme::ChaCha(433, 4, 128)).mask({1, 2, 3});
        EXPECT(false, "dimension mismatch must panic");
    } catch (const Panic&) {}
    if (failures == 0) printf("full_loop.cpp: all scenarios OK\n");
    return failures ? 1 : 0;
}
