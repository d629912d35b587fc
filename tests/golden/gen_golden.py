#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the Python big-int oracle.

The reference (/root/reference) is Rust and cannot be built or imported in this image (no rustc /
cargo, two un-vendored crates), so these vectors do NOT come from running the reference.  They
come from oracle/pyoracle.py - the restatement - with the reference's own test inputs and
expected outputs embedded:

  F0  README walkthrough             README.md:86,105-107,157; docs/simple-cli-example.sh:38-44
  F1  full_loop.rs `simple`          integration-tests/tests/full_loop.rs:29-32,113,148
  F2  full_loop.rs `with_fullmask`   :34-40
  F3  full_loop.rs `with_chachamask` :42-52
  F4  full_loop.rs `with_packedshamir` :54-67
  B*  threshold-secret-sharing 0.2 unit-test vectors [recalled], SURVEY.md Appendix B
  C*  rand 0.3 / RFC 7539 ChaCha vectors, SURVEY.md Appendix C

For F0-F4 the reference asserts only the revealed output (it uses OsRng); here the randomness is
injected (fixed below) so every intermediate stage is pinned as well, in both value modes.

Run from the repo root:  python tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# Provenance vocabulary of every golden file (what pins each part, so a reader need not trace the generator):
#   reference-held   inputs / expected outputs that appear in /root/reference's own tests or README
#   recalled         vectors of an un-vendored crate's unit tests, recalled in SURVEY.md App. B / C - NOT reference output
#   published-RFC    vectors of a published specification
#   oracle-generated produced by this repository's oracle (oracle/pyoracle.py); pins implementations against each other only
#   (reference-generated: only tests/golden/reference_generated.json, made by tests/reference_harness where cargo exists)
PROV_FULL_LOOP = {
    "inputs, aggregation parameters, expected_positive": "reference-held (integration-tests/tests/full_loop.rs:29-67,113,148; README.md:86,105-107,157)",
    "mask_rand, share_rand, clerk_subset": "oracle-generated (injected randomness: the reference draws from OsRng)",
    "stages (every intermediate value, both value modes)": "oracle-generated",
}
PROV_KATS = {
    "B1_recover_polynomial.recalled_*, B2_evaluate_polynomial.recalled_canonical": "recalled (threshold-secret-sharing 0.2 unit tests test_recover_polynomial / test_evaluate_polynomial, SURVEY.md App. B)",
    "B1 / B2 signed and canonical": "oracle-generated (equal to the recalled vectors, signs included)",
    "B3_share, B4_share_matrix_row0, B5_reconstruct": "oracle-generated (FFT form == Lagrange-matrix form; derived in SURVEY.md App. B)",
    "C1_chacha20_zero_key_block0.expected_first4": "published-RFC (RFC 7539 / draft-nir zero-key keystream; also rand's test_rng_true_values [recalled])",
    "C2_masks_seed0, C3_masks_seed1234": "oracle-generated from the RFC-pinned block function and the RECALLED rand 0.3 next_u64 / gen_range rules (SURVEY.md App. C) - parity unpinned by any reference test",
}
PROV_P62 = {"everything": "oracle-generated (62-bit prime scenarios; the reference's own packed-Shamir arithmetic overflows there, SURVEY.md App. A.3)"}
PROV_DRBG = {"cases, call_keys": "oracle-generated (sda-drbg-v1 is the product's own stream layout; the reference uses OsRng)",
             "share_map_cases": "oracle-generated (the systematic CSPRNG share map is the product's own, include/sda_hip.h; each case is "
                                "tied to tss's map - recalled, SURVEY.md App. B - through the implied randomness)",
             "the ChaCha20 block function underneath": "published-RFC (RFC 7539 2.3.2, pinned in tests/test_oracle.py::test_chacha_kats)"}


def scenario(name, source, expected_positive, aggregation, inputs, seed, clerk_subset=None):
    rnd = random.Random(seed)
    msch, ssch = aggregation["masking_scheme"], aggregation["committee_sharing_scheme"]
    dim = aggregation["vector_dimension"]
    q = po.sharing_modulus(ssch)
    gen = po.new_share_generator(ssch)
    k, t = gen.batch_input_size(), gen.rand_per_batch()
    nb = (dim + k - 1) // k
    mask_rand, share_rand = [], []
    for _ in inputs:
        if msch["kind"] == "Full":
            mask_rand.append([rnd.randrange(msch["modulus"]) for _ in range(dim)])
        elif msch["kind"] == "ChaCha":
            mask_rand.append([rnd.getrandbits(32) for _ in range((msch["seed_bitsize"] + 31) // 32)])
        else:
            mask_rand.append([])
        # tss draws from [0, p-1) (Range::new(0, prime - 1)); additive from [0, q)
        hi = q - 1 if ssch["kind"] == "PackedShamir" else q
        share_rand.append([rnd.randrange(hi) for _ in range(nb * t)])
    stages = {}
    for mode in ("rust_signed", "canonical"):
        r = po.full_aggregation(aggregation, inputs, mask_rand, share_rand, clerk_subset, mode)
        assert r["positive"] == expected_positive, (name, mode, r["positive"])
        stages[mode] = r
    return {"name": name, "source": source, "aggregation": aggregation, "inputs": inputs,
            "mask_rand": mask_rand, "share_rand": share_rand, "clerk_subset": clerk_subset,
            "expected_positive": expected_positive, "stages": stages}


def main():
    add433 = dict(kind="Additive", share_count=3, modulus=433)
    agg = lambda mask, share, dim=4: dict(vector_dimension=dim, modulus=433, masking_scheme=mask,   # noqa: E731
                                          committee_sharing_scheme=share)
    two = [[1, 2, 3, 4], [1, 2, 3, 4]]
    fl = "integration-tests/tests/full_loop.rs"
    scenarios = [
        scenario("F0_readme_walkthrough", "README.md:86,105-107,157; docs/simple-cli-example.sh:38-44",
                 [0, 2, 2, 4, 4, 6, 6, 8, 8, 10], agg(dict(kind="None"), add433, 10),
                 [list(range(10)), [0] * 10, [0, 1] * 5], 1000),
        scenario("F1_simple", fl + ":29-32,113,148", [2, 4, 6, 8], agg(dict(kind="None"), add433), two, 1001),
        scenario("F2_with_fullmask", fl + ":34-40", [2, 4, 6, 8], agg(dict(kind="Full", modulus=433), add433), two, 1002),
        scenario("F3_with_chachamask", fl + ":42-52", [2, 4, 6, 8],
                 agg(dict(kind="ChaCha", modulus=433, dimension=4, seed_bitsize=128), add433), two, 1003),
        scenario("F4_with_packedshamir", fl + ":54-67", [2, 4, 6, 8], agg(dict(kind="None"), dict(po.PSS_433)), two, 1004),
        # same shape, a clerk missing (server/src/server.rs:119-120: ready once results >= t+k)
        scenario("F4b_packedshamir_clerk_subset", fl + ":54-67 + server/src/server.rs:119-120", [2, 4, 6, 8],
                 agg(dict(kind="None"), dict(po.PSS_433)), two, 1005, clerk_subset=[0, 1, 2, 3, 4, 5, 7]),
        scenario("F4c_packedshamir_fullmask", fl + ":34-40 x :54-67", [2, 4, 6, 8],
                 agg(dict(kind="Full", modulus=433), dict(po.PSS_433)), two, 1006),
    ]
    with open(os.path.join(OUT, "full_loop.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "provenance": PROV_FULL_LOOP, "scenarios": scenarios}, f, indent=1)

    # ---- KATs of the third-party algorithms ---------------------------------------------------------
    pss = po.PackedSecretSharing(4, 8, 3, 433, 354, 150)
    pss26 = po.PackedSecretSharing(4, 26, 3, 433, 354, 17)
    coeffs = pss.recover_polynomial([1, 2, 3], [8, 8, 8, 8])
    shares = pss.share_fft([1, 2, 3], [8, 8, 8, 8])
    kats = {
        "B1_recover_polynomial": {"values": [0, 1, 2, 3, 8, 8, 8, 8], "prime": 433, "omega": 354,
                                  "signed": coeffs, "canonical": [c % 433 for c in coeffs],
                                  "recalled_signed": [113, -382, -172, 267, -325, 432, 388, -321],
                                  "recalled_canonical": [113, 51, 261, 267, 108, 432, 388, 112]},
        "B2_evaluate_polynomial": {"coefficients_canonical": [c % 433 for c in coeffs], "prime": 433, "omega": 17,
                                   "canonical": [e % 433 for e in pss26.evaluate_polynomial(coeffs + [0] * 19)],
                                   "recalled_canonical": [0, 77, 230, 91, 286, 179, 337, 83, 212, 88, 406, 58, 425, 345,
                                                          350, 336, 430, 404, 51, 60, 305, 395, 84, 156, 160, 112, 422]},
        "B3_share": {"scheme": po.PSS_433, "secrets": [1, 2, 3], "randomness": [8, 8, 8, 8],
                     "canonical": [s % 433 for s in shares], "expected": [91, 337, 88, 425, 336, 51, 395, 160]},
        "B4_share_matrix_row0": {"scheme": po.PSS_433, "row": pss.share_matrix()[0],
                                 "expected": [209, 256, 107, 192, 198, 295, 8]},
        "B5_reconstruct": {"scheme": po.PSS_433, "shares": [s % 433 for s in shares],
                           "index_sets": [[0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 7], [1, 2, 3, 4, 5, 6, 7]],
                           "expected": [1, 2, 3]},
        "C1_chacha20_zero_key_block0": {"words": [hex(w) for w in po.chacha_block(list(po.CHACHA_CONST) + [0] * 12)],
                                        "expected_first4": ["0xade0b876", "0x903df1a0", "0xe56a5d40", "0x28bd8653"]},
        "C2_masks_seed0": {"seed": [0, 0, 0, 0], "modulus": 433, "masks": po.ChaChaMasker(433, 8, 128).expand([0, 0, 0, 0]),
                           "expected": [346, 285, 278, 250, 340, 171, 12, 427]},
        "C3_masks_seed1234": {"seed": [1, 2, 3, 4], "modulus": 433, "masks": po.ChaChaMasker(433, 8, 128).expand([1, 2, 3, 4]),
                              "expected": [59, 358, 179, 210, 379, 368, 395, 356],
                              "first_u64": hex(po.ChaChaRng([1, 2, 3, 4]).next_u64()), "expected_first_u64": "0xea54ec620210af6f"},
    }
    with open(os.path.join(OUT, "kats.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "provenance": PROV_KATS, "kats": kats}, f, indent=1)

    # ---- 62-bit configurations: exact Z_p vectors from the big-int oracle ------
    rnd = random.Random(62)
    P = po.P62
    big = []
    shapes = [("cfg3_k3_t1_n8", 3, 1, 8, 8, 9), ("cfg3ref_k3_t4_n8", 3, 4, 8, 8, 9),
              ("cfg4_k8_t2_n26", 8, 2, 26, 16, 27), ("cfg4ref_k8_t7_n26", 8, 7, 26, 16, 27)]
    for name, k, t, n, o2, o3 in shapes:
        sch = dict(kind="PackedShamir", secret_count=k, share_count=n, privacy_threshold=t, prime_modulus=P,
                   omega_secrets=po.P62_OMEGA[o2], omega_shares=po.P62_OMEGA[o3])
        dim = 3 * k + 2                                   # ragged last batch
        nb = (dim + k - 1) // k
        inputs = [[rnd.randrange(P) for _ in range(dim)] for _ in range(3)]
        inputs[1][0] = -5; inputs[1][1] = P + 7; inputs[2][2] = -(2 ** 62)     # un-range-checked i64 inputs (no i64 overflow in the Rust path)
        share_rand = [[rnd.randrange(P) for _ in range(nb * t)] for _ in inputs]
        a = dict(vector_dimension=dim, modulus=P, masking_scheme=dict(kind="None"), committee_sharing_scheme=sch)
        r = po.full_aggregation(a, inputs, [[]] * 3, share_rand, list(range(n))[::-1][:t + k], "canonical")
        assert r["positive"] == [sum(col) % P for col in zip(*inputs)]
        pss_ = po.PackedSecretSharing(t, n, k, P, sch["omega_secrets"], sch["omega_shares"])
        big.append({"name": name, "aggregation": a, "inputs": inputs, "share_rand": share_rand,
                    "clerk_subset": list(range(n))[::-1][:t + k], "stages": {"canonical": r},
                    "share_matrix_row0": pss_.share_matrix()[0], "is_tss_fft_shape": pss_.is_fft_shape()})
    add = dict(kind="Additive", share_count=3, modulus=P)
    dim = 7
    inputs = [[rnd.randrange(P) for _ in range(dim)] for _ in range(4)]
    inputs[0][0] = -1; inputs[3][6] = 2 ** 62 - 1
    for mk in (dict(kind="None"), dict(kind="Full", modulus=P), dict(kind="ChaCha", modulus=P, dimension=dim, seed_bitsize=128)):
        mask_rand = [([rnd.randrange(P) for _ in range(dim)] if mk["kind"] == "Full" else
                      [rnd.getrandbits(32) for _ in range(4)] if mk["kind"] == "ChaCha" else []) for _ in inputs]
        share_rand = [[rnd.randrange(P) for _ in range(dim * 2)] for _ in inputs]
        a = dict(vector_dimension=dim, modulus=P, masking_scheme=mk, committee_sharing_scheme=add)
        stages = {m: po.full_aggregation(a, inputs, mask_rand, share_rand, None, m) for m in ("rust_signed", "canonical")}
        assert stages["canonical"]["positive"] == [sum(col) % P for col in zip(*inputs)]
        big.append({"name": "cfg2_additive_" + mk["kind"].lower(), "aggregation": a, "inputs": inputs,
                    "mask_rand": mask_rand, "share_rand": share_rand, "clerk_subset": None, "stages": stages})
    with open(os.path.join(OUT, "p62.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "provenance": PROV_P62, "prime": P, "scenarios": big}, f, indent=1)

    # ---- sda-drbg-v1 vectors (product CSPRNG layout; pins the C oracle's copy and the device) --------
    key = bytes(range(32))
    drbg = {"key_hex": key.hex(), "cases": []}
    for (stream, batches, T, m, rounds) in [(0, 19, 1, P, 20), (5, 9, 4, P, 20), (2 ** 40 + 3, 17, 2, 433, 20),
                                            (7, 16, 2, (1 << 61) + 1, 20), (1, 10, 3, P, 12), (1, 10, 3, P, 8),
                                            # the PAIRED rule (moduli <= 0x7F7F7F: one candidate word -> two draws), odd and even T,
                                            # its upper edge and the first modulus above it
                                            (3, 17, 3, 433, 20), (11, 9, 5, 746497, 20), (4, 16, 2, 0x7F7F7F, 20),
                                            (4, 16, 2, 0x7F7F80, 20), (9, 10, 3, 433, 12), (6, 24, 1, 5038849, 8)]:
        drbg["cases"].append({"stream": stream, "batches": batches, "T": T, "modulus": m, "rounds": rounds,
                              "values": po.drbg_fill(key, stream, batches, T, m, rounds)})
    drbg["call_keys"] = [{"call_index": i, "key_hex": po.drbg_call_key(key, i).hex()} for i in (0, 1, 2, (1 << 32) + 5)]
    # ---- the CSPRNG share map (include/sda_hip.h): draws of sda-drbg-v1 -> the shares a generator WITHOUT injected randomness
    # hands out.  Systematic map (every matrix-form kernel): shares 0..t-1 = the draws, the rest by interpolation; each case
    # also carries the randomness tss's own map would need for the same shares (the two describe the same sharings).
    rnd = random.Random(20260929)
    drbg["share_map_cases"] = []
    for (k, t, n, m, w2, w3, stream, dim) in [(3, 1, 8, P, po.P62_OMEGA[8], po.P62_OMEGA[9], 0, 14),
                                              (3, 4, 8, P, po.P62_OMEGA[8], po.P62_OMEGA[9], 3, 10),
                                              (8, 2, 26, P, po.P62_OMEGA[16], po.P62_OMEGA[27], 2 ** 40 + 1, 17),
                                              (8, 7, 26, P, po.P62_OMEGA[16], po.P62_OMEGA[27], 9, 16),
                                              (3, 4, 8, 433, 354, 150, 1, 7)]:
        pss = po.PackedSecretSharing(t, n, k, m, w2, w3)
        B = (dim + k - 1) // k
        secrets = [rnd.randrange(m) for _ in range(dim)]
        draws = po.drbg_fill(key, stream, B, t, m)
        shares = [[] for _ in range(n)]
        implied = []
        for b in range(B):
            batch = secrets[b * k:(b + 1) * k]
            batch += [0] * (k - len(batch))
            d = draws[b * t:(b + 1) * t]
            sh = pss.share_systematic(batch, d)
            imp = pss.implied_tss_randomness(batch, d)
            assert sh == pss.share_lagrange(batch, imp) and sh[:t] == d
            implied += imp
            for j in range(n):
                shares[j].append(sh[j])
        drbg["share_map_cases"].append({"secret_count": k, "privacy_threshold": t, "share_count": n, "modulus": m,
                                        "omega_secrets": w2, "omega_shares": w3, "stream": stream, "secrets": secrets,
                                        "draws": draws, "systematic_shares": shares, "implied_tss_randomness": implied})
    with open(os.path.join(OUT, "drbg.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "provenance": PROV_DRBG, "spec": "sda-drbg-v1 (DESIGN.md)", **drbg}, f, indent=1)
    print("wrote", sorted(x for x in os.listdir(OUT) if x.endswith(".json")))


if __name__ == "__main__":
    main()
