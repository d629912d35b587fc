//! Emits, as one JSON document on stdout, what the crates the reference links compute for the inputs of
//! `oracle_inputs.rs` (generated from this repository's oracle by `gen_inputs.py`):
//!
//!  (i)  `ChaChaRng::from_seed(&seed).gen_range(0_i64, q)` prefixes and the first `next_u64()` - exactly the calls of
//!       client/src/crypto/masking/chacha.rs:36-39 / :67-69 - for several seeds (2, 4 and 8 words) and moduli, one of them with
//!       a 25 % rejection rate so that the zone rule is pinned too;
//!  (i') raw `next_u32()` words after `set_counter` just below 2^32, 2^64 and 2^96 blocks: the carries of the 128-bit block
//!       counter (words 12..15), which `from_seed` streams only reach after 2^35 masks;
//!  (ii) for tss's parameter sets: `share(&secrets)` (fresh OsRng randomness inside tss, as in packed_shamir.rs:42) and
//!       `reconstruct(&subset, &oracle_shares)` (packed_shamir.rs:76) applied to shares THE ORACLE generated - so that
//!       oracle-reconstruct(tss shares) and tss-reconstruct(oracle shares) both have to give back the secrets: cross-
//!       implementation round trips pin the evaluation-point convention (index i <-> omega_shares^(i+1)) without injecting
//!       randomness into tss.
//!
//! Values are printed as tss / rand return them (tss may return negative representatives: Rust's `%`); the consumer
//! (tests/test_oracle.py::test_reference_generated_fixtures) compares modulo the prime.
extern crate rand;
extern crate threshold_secret_sharing as tss;

use rand::{ChaChaRng, Rng, SeedableRng};

mod oracle_inputs;
use oracle_inputs::{CHACHA_CASES, PSS_CASES, RAW_CASES};

fn ints<T: std::fmt::Debug>(v: &[T]) -> String {
    format!("{:?}", v) // "[1, -2, 3]" is valid JSON for integers
}

fn main() {
    let mut chacha: Vec<String> = Vec::new();
    for case in CHACHA_CASES.iter() {
        // chacha.rs:36-39
        let mut rng = ChaChaRng::from_seed(case.seed);
        let masks: Vec<i64> = (0..case.count).map(|_| rng.gen_range(0_i64, case.modulus)).collect();
        let mut fresh = ChaChaRng::from_seed(case.seed);
        let first_u64: u64 = fresh.next_u64();
        let first_u32s: Vec<u32> = {
            let mut r = ChaChaRng::from_seed(case.seed);
            (0..4).map(|_| r.next_u32()).collect()
        };
        chacha.push(format!(
            "{{\"name\": \"{}\", \"seed\": {}, \"modulus\": {}, \"masks\": {}, \"first_next_u64\": {}, \"first_next_u32s\": {}}}",
            case.name, ints(case.seed), case.modulus, ints(&masks), first_u64, ints(&first_u32s)
        ));
    }

    // raw keystream words around the carries of the 128-bit block counter (rand 0.3 ChaChaRng::set_counter): the reference
    // itself always starts at 0 (chacha.rs:36) - this pins the counter layout its long streams would run into
    let mut raw: Vec<String> = Vec::new();
    for case in RAW_CASES.iter() {
        let mut r = ChaChaRng::from_seed(case.seed);
        if case.counter_low != 0 || case.counter_high != 0 {
            r.set_counter(case.counter_low, case.counter_high);
        }
        let words: Vec<u32> = (0..case.words).map(|_| r.next_u32()).collect();
        raw.push(format!(
            "{{\"name\": \"{}\", \"seed\": {}, \"counter_low\": {}, \"counter_high\": {}, \"next_u32s\": {}}}",
            case.name, ints(case.seed), case.counter_low, case.counter_high, ints(&words)
        ));
    }

    let mut pss_out: Vec<String> = Vec::new();
    for case in PSS_CASES.iter() {
        // packed_shamir.rs:14-21
        let pss = tss::packed::PackedSecretSharing {
            threshold: case.threshold,
            share_count: case.share_count,
            secret_count: case.secret_count,
            prime: case.prime,
            omega_secrets: case.omega_secrets,
            omega_shares: case.omega_shares,
        };
        // packed_shamir.rs:42 - randomness drawn inside tss
        let tss_shares: Vec<i64> = pss.share(case.secrets);
        // packed_shamir.rs:76 - tss reconstructs from the ORACLE's shares of the same secrets (subset of clerk indices)
        let picked: Vec<i64> = case.subset.iter().map(|&i| case.oracle_shares[i]).collect();
        let tss_reconstruct_of_oracle_shares: Vec<i64> = pss.reconstruct(case.subset, &picked);
        // and from its own, every clerk
        let all: Vec<usize> = (0..case.share_count).collect();
        let tss_reconstruct_of_tss_shares: Vec<i64> = pss.reconstruct(&all, &tss_shares);
        pss_out.push(format!(
            "{{\"name\": \"{}\", \"threshold\": {}, \"share_count\": {}, \"secret_count\": {}, \"prime\": {}, \"omega_secrets\": {}, \
             \"omega_shares\": {}, \"reconstruct_limit\": {}, \"secrets\": {}, \"subset\": {}, \"tss_shares\": {}, \
             \"tss_reconstruct_of_oracle_shares\": {}, \"tss_reconstruct_of_tss_shares\": {}}}",
            case.name, case.threshold, case.share_count, case.secret_count, case.prime, case.omega_secrets, case.omega_shares,
            pss.reconstruct_limit(), ints(case.secrets), ints(case.subset), ints(&tss_shares),
            ints(&tss_reconstruct_of_oracle_shares), ints(&tss_reconstruct_of_tss_shares)
        ));
    }

    println!("{{");
    println!(" \"provenance\": \"reference-generated: threshold-secret-sharing 0.2 and rand 0.3, the crates client/Cargo.toml:15,18 names, run by tests/reference_harness (commit its Cargo.lock beside this file)\",");
    println!(" \"generator\": \"tests/reference_harness/src/main.rs\",");
    println!(" \"chacha\": [\n  {}\n ],", chacha.join(",\n  "));
    println!(" \"raw\": [\n  {}\n ],", raw.join(",\n  "));
    println!(" \"pss\": [\n  {}\n ]", pss_out.join(",\n  "));
    println!("}}");
}
