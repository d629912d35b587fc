#!/usr/bin/env python3
"""Host-buffer (trait-shaped) entry points: what a Rust caller holding Vec<Vec<i64>> in host memory would see.
PCIe-inclusive by construction; prints one JSON object."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sda_amd import capi, crypto  # noqa: E402

P62 = 4611686006577364993
W = {8: 631229665360524489, 9: 3451275676410824977}
out = {}


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


rng = np.random.default_rng(1)
sch = crypto.PackedShamir(3, 8, 1, P62, W[8], W[9])
dim = 1 << 20
B = (dim + 2) // 3
# clerk: combine P vectors of L = B values held in host memory (clerk.rs:85-86)
P = 2000
shares = rng.integers(0, P62, size=(P, B), dtype=np.int64)
comb = crypto.ShareCombiner(sch)
dt = timed(lambda: comb.combine(shares))
out["clerk_combine_host_2000x349526"] = {"ms": dt * 1e3, "GBps_host_to_result": shares.nbytes / dt / 1e9,
                                         "values_per_s": shares.size / dt}
# participant: generate the 8 share vectors of one 1 Mi vector (participate.rs:75-76)
gen = crypto.ShareGenerator(sch)
sec = rng.integers(0, P62, size=dim, dtype=np.int64)
dt = timed(lambda: gen.generate(sec), reps=5)
out["participant_generate_host_dim1Mi"] = {"ms": dt * 1e3, "elements_per_s": dim / dt}
# recipient: reconstruct from 4 clerk sums (receive.rs:140-144)
rec = crypto.SecretReconstructor(sch, dim)
sums = [(c, rng.integers(0, P62, size=B, dtype=np.int64)) for c in range(4)]
dt = timed(lambda: rec.reconstruct(sums), reps=5)
out["recipient_reconstruct_host_dim1Mi"] = {"ms": dt * 1e3, "secrets_per_s": dim / dt}
print(json.dumps(out, indent=1))
