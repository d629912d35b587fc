#!/usr/bin/env python3
"""bench.py - share-gen + clerk-sum elements/sec (mod q) on MI355X, the metric of BASELINE.json.

One "step" = one pass of the hot path over one tile of synthetic participants resident in HBM:
  share generation (device CSPRNG)  ->  shares materialised in HBM, job-major [n][P_tile][Bs]
  per-clerk modular sum             ->  exact 128-bit accumulators [n][B]
An *element* is one (participant, vector component) pair, so a step processes P_tile * dim elements.
Default workload = BASELINE config 3 (configs[2]): packed Shamir t=1, k=3, n=8, dim 1,048,576, 62-bit
prime, 100k participants = 50 steps of a 2000-participant tile (the configuration the north-star
target is quoted on).  `--workload additive` = config 2 (configs[1]); at N=1 a short run of it is
attached to the JSON line as `additional_workloads`.

N > 1: one process per GPU (torchrun), participants sharded across ranks (weak scaling: the per-GPU
tile is fixed), no collective on the data path, ONE modular reduce of the partial clerk sums over
RCCL at the end of the timed region (sda_amd/distributed.py).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the byte accounting.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P62 = 4611686006577364993
OMEGA = {8: 631229665360524489, 9: 3451275676410824977, 16: 2589100645267092065, 27: 365137883145458390}
SEED = 0x5DA5DA5DA5DA5DA5
KEY = bytes((i * 7 + 1) & 0xFF for i in range(32))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "packed": dict(kind="packed", n=8, k=3, t=1, o2=8, o3=9, participants=100_000,
                   desc="BASELINE config 3: packed Shamir t=1 k=3 n=8, dim 1048576, 62-bit prime, 100k participants"),
    "packed_ref": dict(kind="packed", n=8, k=3, t=4, o2=8, o3=9, participants=100_000,
                       desc="reference-valid tss shape t=4 k=3 n=8 (t+k+1 = 8), dim 1048576, 62-bit prime"),
    "packed26": dict(kind="packed", n=26, k=8, t=2, o2=16, o3=27, participants=1_000_000,
                     desc="BASELINE config 4 shape: packed Shamir t=2 k=8 n=26, dim 1048576, 62-bit prime"),
    "packed26_ref": dict(kind="packed", n=26, k=8, t=7, o2=16, o3=27, participants=1_000_000,
                         desc="reference-valid tss shape t=7 k=8 n=26 (t+k+1 = 16, n+1 = 27), dim 1048576"),
    "packed_k4t3": dict(kind="packed", n=8, k=4, t=3, o2=8, o3=9, participants=100_000,
                        desc="tss-valid shape t=3 k=4 n=8 through the run-time (k, t) dual-role kernel, dim 1048576"),
    "additive": dict(kind="additive", n=3, k=1, t=2, o2=8, o3=9, participants=10_000,
                     desc="BASELINE config 2: additive 3-way, dim 1048576, 62-bit modulus, 10k participants"),
}


def algorithmic_bytes_per_element(n, k):
    """share-gen reads 8 B (secret) and writes 8n/k B; clerk-sum reads 8n/k B."""
    return 8.0 + 8.0 * n / k, 8.0 * n / k


def cpu_baseline(w, dim, budget_s=15.0):
    """The oracle's reference-faithful scalar port (share-gen + clerk-sum), single thread like the
    reference, on a bounded sample of the same workload.  Reported, never the thing shipped."""
    from oracle import coracle
    packed = 1 if w["kind"] == "packed" else 0
    a = (packed, P62, w["n"], w["k"], w["t"], OMEGA[w["o2"]], OMEGA[w["o3"]])
    t0 = time.perf_counter()
    coracle.baseline_pass(*a, 1, dim, 0, SEED, KEY)
    one = time.perf_counter() - t0
    parts = max(1, min(1024, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    done, _ = coracle.baseline_pass(*a, parts, dim, 0, SEED, KEY)
    dt = time.perf_counter() - t0
    res = {"value": done / dt, "unit": "elements/s", "cores": 1, "kind": "port",
           "sample": f"{parts} participants x dim {dim} (share-gen incl. buffered ChaCha20 draws + clerk-sum), "
                     f"oracle/sda_oracle.c single thread, {dt:.1f} s",
           "host_cpus": os.cpu_count()}
    # the same port over participants on many host cores (the reference itself has no threading);
    # each thread owns a participant range and its own clerk sums - the final n x B modular merge is negligible
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 64))
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        per = max(1, int(parts * 0.25))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            outs = list(ex.map(lambda i: coracle.baseline_pass(*a, per, dim, i * per, SEED, KEY)[0], range(threads)))
        dtm = time.perf_counter() - t0
        res["all_cores"] = {"value": sum(outs) / dtm, "unit": "elements/s", "cores": threads,
                            "sample": f"{threads} threads x {per} participants x dim {dim}, {dtm:.1f} s"}
    return res


class Env:
    """process-wide state: device, distributed group, library"""

    def __init__(self):
        import torch
        import torch.distributed as dist
        from sda_amd import capi
        self.torch, self.dist, self.capi = torch, dist, capi
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # SDA_DIST_BACKEND=gloo + SDA_SHARE_GPU=1: a rehearsal of the N > 1 path on a ONE-GPU box - every rank runs its
        # kernels on device 0 and the exchange is staged through host memory (sda_amd.distributed does that for gloo)
        self.backend = os.environ.get("SDA_DIST_BACKEND", "nccl")
        device_index = 0 if os.environ.get("SDA_SHARE_GPU") == "1" else self.local_rank
        torch.cuda.set_device(device_index)
        self.dev = torch.device("cuda", device_index)
        self.use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # launched by torch.distributed.run
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        self.ctl_dev = self.dev if self.backend == "nccl" else torch.device("cpu")   # where the tiny control tensors live
        self.lib = capi.load()
        capi.check(self.lib.sda_set_device(device_index))

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)


def measure_fused(env, name, dim, P, steps, warmup, row_align=16, verify=True):
    """Software-pipelined schedule: ONE dual-role launch per step generates tile i while the clerk sums of
    tile i-1 are accumulated (sda_share_generator_generate_combine_dev); K + 1 launches cover K tiles."""
    torch, dist, capi, lib, dev = env.torch, env.dist, env.capi, env.lib, env.dev
    from sda_amd import crypto
    from sda_amd.distributed import modular_allreduce
    rank, world = env.rank, env.world
    w = WORKLOADS[name]
    n, k, t = w["n"], w["k"], w["t"]
    scheme = (crypto.PackedShamir(k, n, t, P62, OMEGA[w["o2"]], OMEGA[w["o3"]]) if w["kind"] == "packed"
              else crypto.Additive(n, P62))
    B = (dim + k - 1) // k
    Bs = (B + row_align - 1) // row_align * row_align
    gen = crypto.ShareGenerator(scheme)
    gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(scheme)
    secrets = torch.empty((P, dim), dtype=torch.int64, device=dev)
    shares = [torch.empty((n, P, Bs), dtype=torch.int64, device=dev) for _ in range(2)]
    capi.check(lib.sda_fill_synthetic_dev(secrets.data_ptr(), P, dim, dim, rank * P, SEED, P62, None))
    torch.cuda.synchronize(dev)

    def launch(i, total, ev=None):
        """launch i of total+1: generate tile i (if i < total), sum tile i-1 (if i > 0)"""
        cur, prev = shares[i % 2], shares[(i - 1) % 2]
        if ev:
            capi.check(lib.sda_event_record(ev[0], None))
        gen.generate_combine_dev(comb, secrets.data_ptr(), P if i < total else 0, dim, dim, cur.data_ptr(), Bs, P * Bs,
                                 d_prev=prev.data_ptr() if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                 first_participant=(i * world + rank) * P)
        if ev:
            capi.check(lib.sda_event_record(ev[1], None))

    comb.begin_dev(n, B)
    for i in range(warmup + 1):
        launch(i, warmup)
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B)                                     # discard the warm-up contributions
    evs = []
    for _ in range(2 * (steps + 1)):
        e = C.c_void_p()
        capi.check(lib.sda_event_create(C.byref(e)))
        evs.append(e)
    sums = torch.zeros((n, B), dtype=torch.int64, device=dev)
    if env.use_dist:
        modular_allreduce(sums, P62)
        tw = torch.zeros(1, dtype=torch.float64, device=env.ctl_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    env.barrier()
    t0 = time.perf_counter()
    for i in range(steps + 1):
        launch(i, steps, evs[2 * i:2 * i + 2])
    comb.finish_dev(sums.data_ptr())
    total = modular_allreduce(sums, P62) if env.use_dist else sums
    env.barrier()
    dt = time.perf_counter() - t0
    if env.use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=env.ctl_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = C.c_float()
    launch_ms = []
    for i in range(steps + 1):
        capi.check(lib.sda_event_elapsed_ms(evs[2 * i], evs[2 * i + 1], C.byref(ms)))
        launch_ms.append(ms.value)
    all_ms = sum(launch_ms) / len(launch_ms)                 # what rocprofv3 --stats averages (K+1 launches)
    full = launch_ms[1:steps]                                # launches that carry both roles
    full_ms = sum(full) / len(full) if full else None        # needs >= 2 steps
    for e in evs:
        lib.sda_event_destroy(e)
    verified, reveal_ms = None, None
    if verify:
        rec = crypto.SecretReconstructor(scheme, dim)
        out = torch.empty(dim, dtype=torch.int64, device=dev)
        idx = list(range(scheme.reconstruction_threshold()))
        rows = total[:len(idx)].contiguous()
        rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)      # builds the Lagrange matrix once
        torch.cuda.synchronize(dev)
        t_rev = time.perf_counter()
        rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)
        torch.cuda.synchronize(dev)
        reveal_ms = (time.perf_counter() - t_rev) * 1e3
        cs = crypto.ShareCombiner(crypto.Additive(2, P62))
        cs.begin_dev(1, dim)
        for _ in range(steps):
            cs.update_dev(secrets.data_ptr(), 0, P, dim)
        exp = torch.empty(dim, dtype=torch.int64, device=dev)
        cs.finish_dev(exp.data_ptr())
        exp_total = modular_allreduce(exp, P62) if env.use_dist else exp
        torch.cuda.synchronize(dev)
        verified = bool(torch.equal(out, exp_total))
    elements = float(world) * steps * P * dim
    value = elements / dt
    gen_b, comb_b = algorithmic_bytes_per_element(n, k)
    per_launch_bytes = P * dim * (gen_b + comb_b)
    gbs = steps * per_launch_bytes / (sum(launch_ms) * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{name}:tile{P}:dim{dim}", {}).get("fused_bytes_per_launch")
        except Exception:
            traffic = None
    kern = "fused_packed_l31_kernel" if w["kind"] == "packed" else "fused_additive_kernel"
    res = {
        "metric": "share-gen + clerk-sum elements/sec (mod q)", "value": value, "unit": "elements/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": w["desc"], "name": name, "dim": dim, "tile_participants": P,
                   "participants_total": world * steps * P, "share_count": n, "secret_count": k,
                   "privacy_threshold": t, "modulus": P62, "randomness": f"on-device ChaCha{os.environ.get('SDA_DRBG_ROUNDS', '20')} (sda-drbg-v1)",
                   "row_stride_elements": Bs,
                   "schedule": "dual-role launch: share-gen of tile i and clerk-sum of tile i-1 interleaved in one grid "
                               "(shares materialised in HBM by one launch, read back by the next); K+1 launches for K tiles",
                   "parallelism": f"participants sharded x{world}, one modular reduce at the end"},
        "roofline": {"bound": "hbm", "kernel": kern, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": all_ms,
                     "launches": steps + 1, "both_roles_launch_ms": full_ms,
                     "first_launch_ms_share_gen_only": launch_ms[0], "last_launch_ms_clerk_sum_only": launch_ms[-1],
                     "note": "one launch = share-gen of a tile (8 + 8n/k B/element) + clerk-sum of the previous tile "
                             "(8n/k B/element); K tiles take K+1 launches (the first only generates, the last only "
                             "sums), achieved = K x algorithmic_bytes_per_launch / sum of the K+1 launch durations"},
        "path_roofline": {"bytes_per_element": gen_b + comb_b,
                          "achieved_GBps": value / world * (gen_b + comb_b) / 1e9,
                          "frac_of_hbm_peak": value / world * (gen_b + comb_b) / 1e9 / HBM_PEAK_GBS},
        "verified_reconstruct_equals_sum": verified,
        "reveal": None if reveal_ms is None else {
            "ms": reveal_ms, "secrets_per_s": dim / (reveal_ms * 1e-3),
            "note": "Lagrange reconstruction of the dim secrets from t+k clerk sums (receive.rs:140-152), host-timed "
                    "around one reconstruct_dev call, outside the timed region"},
    }
    del secrets, shares, sums, total
    torch.cuda.empty_cache()
    return res


def measure(env, name, dim, P, steps, warmup, row_align=16, overlap=0, verify=True):
    """K timed steps of workload `name`; returns the metric dict (valid on every rank)."""
    torch, dist, capi, lib, dev = env.torch, env.dist, env.capi, env.lib, env.dev
    from sda_amd import crypto
    from sda_amd.distributed import modular_allreduce
    rank, world = env.rank, env.world
    w = WORKLOADS[name]
    n, k, t = w["n"], w["k"], w["t"]
    if w["kind"] == "packed":
        scheme = crypto.PackedShamir(k, n, t, P62, OMEGA[w["o2"]], OMEGA[w["o3"]])
    else:
        scheme = crypto.Additive(n, P62)
    B = (dim + k - 1) // k
    Bs = (B + row_align - 1) // row_align * row_align      # 128-byte aligned rows (row_align = 16 elements)

    gen = crypto.ShareGenerator(scheme)
    gen.set_drbg_key(KEY)
    comb = crypto.ShareCombiner(scheme)
    if overlap:
        comb.set_residency(2)            # 8 waves per CU saturate HBM; the rest stay with share generation

    # resident tile: secrets [P][dim], shares job-major [n][P][Bs]  (server snapshot layout, stores.rs:86-101)
    nbuf = 2 if overlap else 1
    secrets = torch.empty((P, dim), dtype=torch.int64, device=dev)
    shares = [torch.empty((n, P, Bs), dtype=torch.int64, device=dev) for _ in range(nbuf)]
    if overlap:
        s_gen, s_comb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    else:
        s_gen = s_comb = torch.cuda.current_stream(dev)
    h_gen, h_comb = s_gen.cuda_stream or None, s_comb.cuda_stream or None
    capi.check(lib.sda_fill_synthetic_dev(secrets.data_ptr(), P, dim, dim, rank * P, SEED, P62, None))
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B, h_comb or 0)
    gen_done = [torch.cuda.Event() for _ in range(nbuf)]
    comb_done = [torch.cuda.Event() for _ in range(nbuf)]

    def step(i, slot, evs=None):
        """share-gen of tile i into shares[slot] on s_gen; clerk-sum of it on s_comb"""
        first = (i * world + rank) * P                      # participant ids of this tile (CSPRNG stream ids)
        buf = shares[slot]
        if overlap:
            s_gen.wait_event(comb_done[slot])               # the previous reader of this buffer is done
        if evs:
            capi.check(lib.sda_event_record(evs[0], h_gen))
        gen.generate_batch_dev(secrets.data_ptr(), P, dim, dim, buf.data_ptr(), Bs, P * Bs,
                               first_participant=first, stream=h_gen or 0)
        if evs:
            capi.check(lib.sda_event_record(evs[1], h_gen))
        if overlap:
            gen_done[slot].record(s_gen)
            s_comb.wait_event(gen_done[slot])
        if evs:
            capi.check(lib.sda_event_record(evs[2], h_comb))
        comb.update_dev(buf.data_ptr(), P * Bs, P, Bs, stream=h_comb or 0)
        if evs:
            capi.check(lib.sda_event_record(evs[3], h_comb))
        if overlap:
            comb_done[slot].record(s_comb)

    for e in comb_done:
        e.record(s_comb)
    for i in range(warmup):
        step(-1 - i, i % nbuf)
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B, h_comb or 0)                       # discard the warm-up contributions
    for e in comb_done:
        e.record(s_comb)

    # per-kernel HIP events on the streams the kernels are launched on (4 per step)
    evs = []
    for _ in range(4 * steps):
        e = C.c_void_p()
        capi.check(lib.sda_event_create(C.byref(e)))
        evs.append(e)

    sums = torch.zeros((n, B), dtype=torch.int64, device=dev)
    if env.use_dist:
        # RCCL connects lazily on the first collective of each kind: do that outside the timed region,
        # with the real message sizes
        with torch.cuda.stream(s_comb):
            modular_allreduce(sums, P62)
        tw = torch.zeros(1, dtype=torch.float64, device=env.ctl_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    env.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, i % nbuf, evs[4 * i:4 * i + 4])
    comb.finish_dev(sums.data_ptr(), h_comb or 0)
    with torch.cuda.stream(s_comb):
        total = modular_allreduce(sums, P62) if env.use_dist else sums     # X1: the only exchange step
    env.barrier()
    dt = time.perf_counter() - t0
    if env.use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=env.ctl_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    gen_ms = comb_ms = 0.0
    ms = C.c_float()
    for i in range(steps):
        capi.check(lib.sda_event_elapsed_ms(evs[4 * i], evs[4 * i + 1], C.byref(ms))); gen_ms += ms.value
        capi.check(lib.sda_event_elapsed_ms(evs[4 * i + 2], evs[4 * i + 3], C.byref(ms))); comb_ms += ms.value
    gen_ms /= steps
    comb_ms /= steps
    for e in evs:
        lib.sda_event_destroy(e)

    # size-independent check of the full result: reconstruct(clerk sums) == K * world * (sum of the tile's
    # secrets) mod p -- every step re-shares the same resident tile with fresh randomness
    verified, reveal_ms = None, None
    if verify:
        rec = crypto.SecretReconstructor(scheme, dim)
        out = torch.empty(dim, dtype=torch.int64, device=dev)
        idx = list(range(scheme.reconstruction_threshold()))
        rows = total[:len(idx)].contiguous()
        rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)      # builds the Lagrange matrix once
        torch.cuda.synchronize(dev)
        t_rev = time.perf_counter()
        rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)
        torch.cuda.synchronize(dev)
        reveal_ms = (time.perf_counter() - t_rev) * 1e3
        cs = crypto.ShareCombiner(crypto.Additive(2, P62))   # expected: column sums of the secrets tile, K times
        cs.begin_dev(1, dim)
        for _ in range(steps):
            cs.update_dev(secrets.data_ptr(), 0, P, dim)
        exp = torch.empty(dim, dtype=torch.int64, device=dev)
        cs.finish_dev(exp.data_ptr())
        exp_total = modular_allreduce(exp, P62) if env.use_dist else exp
        torch.cuda.synchronize(dev)
        verified = bool(torch.equal(out, exp_total))

    elements = float(world) * steps * P * dim
    value = elements / dt
    gen_b, comb_b = algorithmic_bytes_per_element(n, k)
    per_launch = P * dim
    gen_gbs = per_launch * gen_b / (gen_ms * 1e-3) / 1e9
    comb_gbs = per_launch * comb_b / (comb_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    dominant_gen = gen_ms >= comb_ms
    if os.path.exists(tpath):
        try:
            tr = json.load(open(tpath)).get(f"{name}:tile{P}:dim{dim}", {})
            traffic = tr.get("gen_bytes_per_launch" if dominant_gen else "comb_bytes_per_launch")
        except Exception:
            traffic = None
    gen_kernel = ("packed_gen_l31_kernel" if w["kind"] == "packed" else "additive_gen_kernel")
    res = {
        "metric": "share-gen + clerk-sum elements/sec (mod q)", "value": value, "unit": "elements/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": w["desc"], "name": name, "dim": dim, "tile_participants": P,
                   "participants_total": world * steps * P, "share_count": n, "secret_count": k,
                   "privacy_threshold": t, "modulus": P62, "randomness": f"on-device ChaCha{os.environ.get('SDA_DRBG_ROUNDS', '20')} (sda-drbg-v1)",
                   "row_stride_elements": Bs,
                   "schedule": ("share-gen(i+1) overlapped with clerk-sum(i) on two streams, double-buffered shares"
                                if overlap else "one stream, serial"),
                   "parallelism": f"participants sharded x{world}, one modular reduce at the end"},
        "roofline": {"bound": "hbm", "kernel": gen_kernel if dominant_gen else "combine_update_kernel",
                     "achieved": gen_gbs if dominant_gen else comb_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (gen_gbs if dominant_gen else comb_gbs) / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": per_launch * (gen_b if dominant_gen else comb_b),
                     "avg_launch_ms": gen_ms if dominant_gen else comb_ms,
                     "note": "the share-gen kernel is VALU-bound as measured (SQ PMC: VALU active 94 %), see DESIGN.md"},
        "kernels": {"share_gen": {"avg_ms": gen_ms, "bytes_per_element": gen_b, "GBps": gen_gbs,
                                  "frac_of_hbm_peak": gen_gbs / HBM_PEAK_GBS},
                    "clerk_sum": {"avg_ms": comb_ms, "bytes_per_element": comb_b, "GBps": comb_gbs,
                                  "frac_of_hbm_peak": comb_gbs / HBM_PEAK_GBS}},
        "path_roofline": {"bytes_per_element": gen_b + comb_b,
                          "achieved_GBps": value / world * (gen_b + comb_b) / 1e9,
                          "frac_of_hbm_peak": value / world * (gen_b + comb_b) / 1e9 / HBM_PEAK_GBS},
        "verified_reconstruct_equals_sum": verified,
        "reveal": None if reveal_ms is None else {
            "ms": reveal_ms, "secrets_per_s": dim / (reveal_ms * 1e-3),
            "note": "Lagrange reconstruction of the dim secrets from t+k clerk sums (receive.rs:140-152), host-timed "
                    "around one reconstruct_dev call, outside the timed region"},
    }
    del secrets, shares, sums, total
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="packed", choices=sorted(WORKLOADS))
    ap.add_argument("--dim", type=int, default=1 << 20)
    ap.add_argument("--tile", type=int, default=2000, help="participants per step and per GPU")
    ap.add_argument("--row-align", type=int, default=16, help="pad share rows to a multiple of this many elements")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: share-gen of tile i+1 runs concurrently with clerk-sum of tile i (two streams, "
                         "double-buffered shares, clerk-sum capped at 2 workgroups per CU); 0: one stream, serial")
    ap.add_argument("--schedule", default="fused", choices=["serial", "fused"],
                    help="serial: share-gen launch then clerk-sum launch per tile; fused: one dual-role launch per "
                         "step (tile i generated while tile i-1 is summed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-additional", action="store_true", help="skip the short config-2 (additive) run")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    env = Env()
    def run(name, steps, warmup):
        if args.schedule == "fused" and not args.overlap:
            return measure_fused(env, name, args.dim, args.tile, steps, warmup, args.row_align, verify=not args.no_verify)
        return measure(env, name, args.dim, args.tile, steps, warmup, args.row_align, args.overlap, verify=not args.no_verify)

    line = run(args.workload, args.steps, args.warmup)
    if env.world == 1 and not args.no_additional and args.workload == "packed":
        # BASELINE config 2 (additive 3-way, 10k participants = 5 steps of the 2000-participant tile)
        add = run("additive", 5, 2)
        line["additional_workloads"] = {"additive": {k: add[k] for k in ("value", "unit", "ms_per_step", "config", "kernels",
                                                                          "roofline", "path_roofline",
                                                                          "verified_reconstruct_equals_sum", "reveal") if k in add}}
    if env.rank == 0:
        if not args.no_cpu_baseline and env.world == 1:
            line["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload], args.dim)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if env.use_dist:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
