// sda_crypto.hpp - C++ host-side mirror of the reference's `client::crypto` sharing / masking
// interface over the C ABI (include/sda_hip.h).  The reference is compiled code (Rust); its toolchain
// is absent from this image, so the host side above the C ABI is written in C++ with the same names,
// argument meaning and error behaviour:
//
//   reference (client/src/crypto)                         here
//   ------------------------------------------------      -------------------------------------------
//   trait ShareGenerator        sharing/mod.rs:14-17      struct ShareGenerator       ::generate
//   trait ShareCombiner         sharing/mod.rs:23-25      struct ShareCombiner        ::combine
//   trait SecretReconstructor   sharing/mod.rs:31-33      struct SecretReconstructor  ::reconstruct
//   trait SecretMasker          masking/mod.rs:13-15      struct SecretMasker         ::mask
//   trait MaskCombiner          masking/mod.rs:21-23      struct MaskCombiner         ::combine
//   trait SecretUnmasker        masking/mod.rs:29-31      struct SecretUnmasker       ::unmask
//   CryptoModule::new_* factories (sharing/mod.rs:35-96, masking/mod.rs:33-94)   CryptoModule::new_*
//   SdaClientResult<T> Err("...")                         throws SdaClientError("...") (same message)
//   assert!/assert_eq! panics of the masking traits       throws Panic
//
// Header-only; link against libsda_hip.so.  All arithmetic happens on the GPU behind the C ABI.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sda_hip.h"

namespace sda_client {

using Secret = int64_t;        // client/src/crypto/mod.rs:33-36
using Mask = int64_t;
using MaskedSecret = int64_t;
using Share = int64_t;

struct SdaClientError : std::runtime_error {        // error_chain string errors (client/src/errors.rs)
    int code;
    SdaClientError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
struct Panic : std::logic_error {                   // assert!/assert_eq! in the reference
    using std::logic_error::logic_error;
};

namespace detail {
inline void check(int status) {
    if (status == SDA_OK) return;
    std::string msg = sda_last_error();
    if (msg.empty()) msg = sda_strerror(status);
    if (status == SDA_ERR_ASSERTION) throw Panic(msg);
    throw SdaClientError(status, msg);
}
inline std::vector<const int64_t*> row_ptrs(const std::vector<std::vector<int64_t>>& rows, std::vector<size_t>& lens) {
    std::vector<const int64_t*> p(rows.size());
    lens.resize(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) { p[i] = rows[i].data(); lens[i] = rows[i].size(); }
    return p;
}
}  // namespace detail

// ---- protocol/src/crypto.rs:79-155 -------------------------------------------------------------
struct LinearSecretSharingScheme {
    sda_sharing_scheme_t c{};
    static LinearSecretSharingScheme Additive(size_t share_count, int64_t modulus) {
        LinearSecretSharingScheme s;
        s.c.kind = SDA_SHARING_ADDITIVE; s.c.share_count = share_count; s.c.modulus = modulus;
        return s;
    }
    static LinearSecretSharingScheme PackedShamir(size_t secret_count, size_t share_count, size_t privacy_threshold,
                                                  int64_t prime_modulus, int64_t omega_secrets, int64_t omega_shares) {
        LinearSecretSharingScheme s;
        s.c.kind = SDA_SHARING_PACKED_SHAMIR; s.c.share_count = share_count; s.c.modulus = prime_modulus;
        s.c.secret_count = secret_count; s.c.privacy_threshold = privacy_threshold;
        s.c.omega_secrets = omega_secrets; s.c.omega_shares = omega_shares;
        return s;
    }
    size_t input_size() const { return sda_scheme_input_size(&c); }
    size_t output_size() const { return sda_scheme_output_size(&c); }
    size_t privacy_threshold() const { return sda_scheme_privacy_threshold(&c); }
    size_t reconstruction_threshold() const { return sda_scheme_reconstruction_threshold(&c); }
};

// ---- protocol/src/crypto.rs:43-75 --------------------------------------------------------------
struct LinearMaskingScheme {
    sda_masking_scheme_t c{};
    static LinearMaskingScheme None() { LinearMaskingScheme s; s.c.kind = SDA_MASKING_NONE; return s; }
    static LinearMaskingScheme Full(int64_t modulus) {
        LinearMaskingScheme s; s.c.kind = SDA_MASKING_FULL; s.c.modulus = modulus; return s;
    }
    static LinearMaskingScheme ChaCha(int64_t modulus, size_t dimension, size_t seed_bitsize) {
        LinearMaskingScheme s;
        s.c.kind = SDA_MASKING_CHACHA; s.c.modulus = modulus; s.c.dimension = dimension; s.c.seed_bitsize = seed_bitsize;
        return s;
    }
    bool has_mask() const { return sda_masking_has_mask(&c) != 0; }
};

// ---- sharing traits -----------------------------------------------------------------------------
struct ShareGenerator {
    sda_share_generator_t* h = nullptr;
    explicit ShareGenerator(const LinearSecretSharingScheme& s) { detail::check(sda_share_generator_new(&s.c, &h)); }
    // SDA_VALUES_RUST_SIGNED: the reference's own signed representatives (sda_hip.h "value representation")
    void set_value_mode(int mode) { detail::check(sda_share_generator_set_value_mode(h, mode)); }
    // which parametrisation `generate` WITHOUT injected randomness uses (sda_hip.h "CSPRNG share map"): SDA_SHARE_MAP_SYSTEMATIC
    // (the t draws of a batch are its shares 0..t-1; default of the matrix-form kernels) or SDA_SHARE_MAP_TSS_NODES
    int csprng_share_map() const { return sda_share_generator_csprng_share_map(h); }
    void set_csprng_share_map(int map) { detail::check(sda_share_generator_set_csprng_share_map(h, map)); }
    ~ShareGenerator() { sda_share_generator_free(h); }
    ShareGenerator(const ShareGenerator&) = delete;
    /// generate(&mut self, secrets) -> Vec<Vec<Share>>: outer index = clerk (batched.rs:46-48)
    std::vector<std::vector<Share>> generate(const std::vector<Secret>& secrets, const std::vector<int64_t>* rand = nullptr) {
        const size_t n = sda_share_generator_share_count(h), B = sda_share_generator_batch_count(h, secrets.size());
        std::vector<Share> flat(n * B);
        detail::check(sda_share_generator_generate(h, secrets.data(), secrets.size(), rand ? rand->data() : nullptr,
                                                   rand ? rand->size() : 0, flat.data(), flat.size()));
        std::vector<std::vector<Share>> out(n);
        for (size_t j = 0; j < n; ++j) out[j].assign(flat.begin() + j * B, flat.begin() + (j + 1) * B);
        return out;
    }
};

struct ShareCombiner {
    sda_share_combiner_t* h = nullptr;
    explicit ShareCombiner(const LinearSecretSharingScheme& s) { detail::check(sda_share_combiner_new(&s.c, &h)); }
    // SDA_VALUES_RUST_SIGNED: the reference's own signed representatives (sda_hip.h "value representation")
    void set_value_mode(int mode) { detail::check(sda_share_combiner_set_value_mode(h, mode)); }
    ~ShareCombiner() { sda_share_combiner_free(h); }
    ShareCombiner(const ShareCombiner&) = delete;
    std::vector<Share> combine(const std::vector<std::vector<Share>>& shares) const {
        std::vector<size_t> lens;
        auto ptrs = detail::row_ptrs(shares, lens);
        std::vector<Share> out(shares.empty() ? 0 : shares[0].size());
        size_t n_out = 0;
        detail::check(sda_share_combiner_combine(h, ptrs.data(), lens.data(), shares.size(), out.data(), out.size(), &n_out));
        out.resize(n_out);
        return out;
    }
};

// ---- share-vector wire codec (sodium.rs:36-41 `share.encode_var`, :83-89 `Share::decode_var` until EOF) ----------
struct ShareCodec {
    sda_varint_codec_t* h = nullptr;
    ShareCodec() { detail::check(sda_varint_codec_new(&h)); }
    ~ShareCodec() { sda_varint_codec_free(h); }
    ShareCodec(const ShareCodec&) = delete;
    std::vector<uint8_t> encode(const std::vector<Share>& shares) const {
        std::vector<uint8_t> out(sda_varint_max_encoded_size(shares.size()) + 1);
        size_t n = 0;
        detail::check(sda_varint_encode(h, shares.data(), shares.size(), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
    std::vector<Share> decode(const std::vector<uint8_t>& raw) const {
        std::vector<Share> out(raw.size() + 1);
        size_t n = 0;
        detail::check(sda_varint_decode(h, raw.data(), raw.size(), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
};

/// clerk.rs:78-86 as a stream (the FIXME at :71-72): begin(dimension); add(payload) per opened sealed box; finish()
struct StreamingShareCombiner {
    sda_share_combiner_t* h = nullptr;
    size_t dimension = 0;
    explicit StreamingShareCombiner(const LinearSecretSharingScheme& s) { detail::check(sda_share_combiner_new(&s.c, &h)); }
    ~StreamingShareCombiner() { sda_share_combiner_free(h); }
    StreamingShareCombiner(const StreamingShareCombiner&) = delete;
    void begin(size_t dim) { dimension = dim; detail::check(sda_share_combiner_begin(h, dim)); }
    void add(const ShareCodec& codec, const std::vector<uint8_t>& payload) {
        detail::check(sda_share_combiner_update_varint(h, codec.h, payload.data(), payload.size()));
    }
    std::vector<Share> finish() {
        std::vector<Share> out(dimension);
        detail::check(sda_share_combiner_finish(h, out.data()));
        return out;
    }
};

// client/src/crypto/encryption/sodium.rs:33-46, :72-92: ShareEncryptor::encrypt = varint encode +
// sealedbox::seal (:43); ShareDecryptor::decrypt = sealedbox::open (:78; Err("Sodium decryption failure") :80) + decode
using EncryptionKey = std::vector<uint8_t>;         // 32 bytes, EncryptionKey::Sodium
using DecryptionKey = std::vector<uint8_t>;         // 32 bytes, DecryptionKey::Sodium
using Encryption = std::vector<uint8_t>;            // Encryption::Sodium(Binary)
struct SealedBox {
    sda_sealedbox_t* h = nullptr;
    SealedBox() { detail::check(sda_sealedbox_new(&h)); }
    ~SealedBox() { sda_sealedbox_free(h); }
    SealedBox(const SealedBox&) = delete;
    SealedBox& operator=(const SealedBox&) = delete;
    EncryptionKey public_key(const DecryptionKey& sk) {            // X25519(sk, 9), for tests and tools
        EncryptionKey pk(32);
        detail::check(sda_sealedbox_public_key(h, sk.data(), pk.data()));
        return pk;
    }
    Encryption seal(const std::vector<uint8_t>& m, const EncryptionKey& pk, const uint8_t* esk = nullptr) {
        Encryption out(m.size() + SDA_SEALBYTES);
        detail::check(sda_sealedbox_seal(h, pk.data(), esk, m.data(), m.size(), out.data(), out.size()));
        return out;
    }
    std::vector<uint8_t> open(const Encryption& c, const EncryptionKey& pk, const DecryptionKey& sk) {
        std::vector<uint8_t> out(c.size() + 1);
        size_t n = 0;
        detail::check(sda_sealedbox_open(h, pk.data(), sk.data(), c.data(), c.size(), out.data(), out.size(), &n));
        out.resize(n);
        return out;
    }
};
struct ShareEncryptor {
    EncryptionKey pk; SealedBox box; ShareCodec codec;
    explicit ShareEncryptor(EncryptionKey k) : pk(std::move(k)) {}
    Encryption encrypt(const std::vector<Share>& shares) { return box.seal(codec.encode(shares), pk); }
};
struct ShareDecryptor {
    EncryptionKey pk; DecryptionKey sk; SealedBox box; ShareCodec codec;
    ShareDecryptor(EncryptionKey p, DecryptionKey s) : pk(std::move(p)), sk(std::move(s)) {}
    std::vector<Share> decrypt(const Encryption& e) { return codec.decode(box.open(e, pk, sk)); }
};

// The clerking job as one binary blob: what server/src/stores.rs:86-101 would emit instead of
// Vec<Vec<Encryption>>; layout in sda_hip.h ("SDAJOBv1").
struct JobContainer {
    std::vector<uint8_t> blob;
    sda_job_layout_t layout{};
    static JobContainer build(uint32_t kind, const std::vector<std::vector<uint8_t>>& payloads) {
        size_t longest = 0;
        for (const auto& p : payloads) longest = p.size() > longest ? p.size() : longest;
        JobContainer j;
        const size_t slot = sda_job_slot_size(longest);
        const size_t size = sda_job_container_size(payloads.size(), slot);
        j.blob.resize(size < 64 ? 64 : size);
        detail::check(sda_job_container_init(j.blob.data(), j.blob.size(), kind, payloads.size(), slot, &j.layout));
        for (size_t r = 0; r < payloads.size(); ++r)
            detail::check(sda_job_container_set_row(j.blob.data(), j.blob.size(), r, payloads[r].data(), payloads[r].size()));
        return j;
    }
    static JobContainer parse(std::vector<uint8_t> bytes) {
        JobContainer j;
        j.blob = std::move(bytes);
        detail::check(sda_job_container_parse(j.blob.data(), j.blob.size(), &j.layout));
        return j;
    }
    std::vector<uint8_t> row(size_t r) const {
        const uint8_t* p = nullptr;
        size_t n = 0;
        detail::check(sda_job_container_get_row(blob.data(), blob.size(), r, &p, &n));
        return std::vector<uint8_t>(p, p + n);
    }
};

struct SecretReconstructor {
    sda_secret_reconstructor_t* h = nullptr;
    size_t dimension;
    SecretReconstructor(const LinearSecretSharingScheme& s, size_t dim) : dimension(dim) {
        detail::check(sda_secret_reconstructor_new(&s.c, dim, &h));
    }
    ~SecretReconstructor() { sda_secret_reconstructor_free(h); }
    void set_value_mode(int mode) { detail::check(sda_secret_reconstructor_set_value_mode(h, mode)); }
    SecretReconstructor(const SecretReconstructor&) = delete;
    std::vector<Secret> reconstruct(const std::vector<std::pair<size_t, std::vector<Share>>>& indexed_shares) const {
        std::vector<size_t> idx(indexed_shares.size()), lens(indexed_shares.size());
        std::vector<const int64_t*> ptrs(indexed_shares.size());
        size_t cap = dimension;
        for (size_t i = 0; i < indexed_shares.size(); ++i) {
            idx[i] = indexed_shares[i].first; ptrs[i] = indexed_shares[i].second.data(); lens[i] = indexed_shares[i].second.size();
            if (lens[i] > cap) cap = lens[i];
        }
        std::vector<Secret> out(cap);
        size_t n_out = 0;
        detail::check(sda_secret_reconstructor_reconstruct(h, idx.data(), ptrs.data(), lens.data(), idx.size(), out.data(), cap, &n_out));
        out.resize(n_out);
        return out;
    }
};

// ---- masking traits -----------------------------------------------------------------------------
struct SecretMasker {
    sda_secret_masker_t* h = nullptr;
    explicit SecretMasker(const LinearMaskingScheme& s) { detail::check(sda_secret_masker_new(&s.c, &h)); }
    // SDA_VALUES_RUST_SIGNED: the reference's own signed representatives (sda_hip.h "value representation")
    void set_value_mode(int mode) { detail::check(sda_secret_masker_set_value_mode(h, mode)); }
    ~SecretMasker() { sda_secret_masker_free(h); }
    SecretMasker(const SecretMasker&) = delete;
    std::pair<std::vector<Mask>, std::vector<MaskedSecret>> mask(const std::vector<Secret>& secrets,
                                                                 const std::vector<int64_t>* rand = nullptr) {
        std::vector<Mask> m(sda_secret_masker_mask_len(h, secrets.size()));
        std::vector<MaskedSecret> ms(secrets.size());
        size_t n_mask = 0;
        detail::check(sda_secret_masker_mask(h, secrets.data(), secrets.size(), rand ? rand->data() : nullptr,
                                             rand ? rand->size() : 0, m.data(), m.size(), &n_mask, ms.data()));
        m.resize(n_mask);
        return {m, ms};
    }
};

struct MaskCombiner {
    sda_mask_combiner_t* h = nullptr;
    size_t chacha_dimension = 0;
    explicit MaskCombiner(const LinearMaskingScheme& s) {
        if (s.c.kind == SDA_MASKING_CHACHA) chacha_dimension = s.c.dimension;
        detail::check(sda_mask_combiner_new(&s.c, &h));
    }
    ~MaskCombiner() { sda_mask_combiner_free(h); }
    void set_value_mode(int mode) { detail::check(sda_mask_combiner_set_value_mode(h, mode)); }
    MaskCombiner(const MaskCombiner&) = delete;
    std::vector<Mask> combine(const std::vector<std::vector<Mask>>& masks) const {
        std::vector<size_t> lens;
        auto ptrs = detail::row_ptrs(masks, lens);
        std::vector<Mask> out(chacha_dimension ? chacha_dimension : (masks.empty() ? 0 : masks[0].size()));
        size_t n_out = 0;
        detail::check(sda_mask_combiner_combine(h, ptrs.data(), lens.data(), masks.size(), out.data(), out.size(), &n_out));
        out.resize(n_out);
        return out;
    }
};

struct SecretUnmasker {
    sda_secret_unmasker_t* h = nullptr;
    explicit SecretUnmasker(const LinearMaskingScheme& s) { detail::check(sda_secret_unmasker_new(&s.c, &h)); }
    // SDA_VALUES_RUST_SIGNED: the reference's own signed representatives (sda_hip.h "value representation")
    void set_value_mode(int mode) { detail::check(sda_secret_unmasker_set_value_mode(h, mode)); }
    ~SecretUnmasker() { sda_secret_unmasker_free(h); }
    SecretUnmasker(const SecretUnmasker&) = delete;
    std::vector<Secret> unmask(const std::pair<std::vector<Mask>, std::vector<MaskedSecret>>& values) const {
        std::vector<Secret> out(values.second.size());
        detail::check(sda_secret_unmasker_unmask(h, values.first.data(), values.first.size(), values.second.data(),
                                                 values.second.size(), out.data()));
        return out;
    }
};

// ---- client/src/crypto/mod.rs:58-66 + the six *Construction traits ---------------------------------
struct CryptoModule {
    std::unique_ptr<ShareGenerator> new_share_generator(const LinearSecretSharingScheme& s) const { return std::make_unique<ShareGenerator>(s); }
    std::unique_ptr<ShareCombiner> new_share_combiner(const LinearSecretSharingScheme& s) const { return std::make_unique<ShareCombiner>(s); }
    std::unique_ptr<SecretReconstructor> new_secret_reconstructor(const LinearSecretSharingScheme& s, size_t dimension) const {
        return std::make_unique<SecretReconstructor>(s, dimension);
    }
    std::unique_ptr<SecretMasker> new_secret_masker(const LinearMaskingScheme& s) const { return std::make_unique<SecretMasker>(s); }
    std::unique_ptr<MaskCombiner> new_mask_combiner(const LinearMaskingScheme& s) const { return std::make_unique<MaskCombiner>(s); }
    std::unique_ptr<SecretUnmasker> new_secret_unmasker(const LinearMaskingScheme& s) const { return std::make_unique<SecretUnmasker>(s); }
};

// ---- client/src/receive.rs:7-21 ---------------------------------------------------------------------
struct RecipientOutput {
    int64_t modulus;
    std::vector<int64_t> values;
    RecipientOutput positive() const {
        RecipientOutput o{modulus, std::vector<int64_t>(values.size())};
        detail::check(sda_positive(values.data(), values.size(), modulus, o.values.data()));
        return o;
    }
};

}  // namespace sda_client
