#!/usr/bin/env python3
"""bench.py - share-gen + clerk-sum elements/sec (mod q) on MI355X, the metric of BASELINE.json.

One "step" = one pass of the hot path over one tile of synthetic participants resident in HBM:
  share generation (device CSPRNG)  ->  shares materialised in HBM, job-major [n][P_tile][Bs]
  per-clerk modular sum             ->  exact 128-bit accumulators [n][B]
An *element* is one (participant, vector component) pair, so a step processes P_tile * dim elements.
Default workload = BASELINE config 3 (configs[2]): packed Shamir t=1, k=3, n=8, dim 1,048,576, 62-bit
prime, 100k participants (the configuration the north-star target is quoted on).  The K timed steps
cover ALL of them: a step = 100000 / K participants, issued as resident sub-tiles of <= 2500
(50 steps x 2000; 20 steps x 2 x 2500); `config.workload` is derived from what was processed and
`config.inputs` says what was shared (default: one resident tile replayed; --inputs distinct: 100k different
participants).  `--workload additive` = config 2 (configs[1]); at N=1 short runs of it, of config 5's dimension (with
the reveal), of config 4's shape and of tss's PSS_155_728_100 (transform kernel) are attached to the JSON line as
`additional_workloads`.

N > 1: one process per GPU (torchrun), participants sharded across ranks, no collective on the data path,
ONE modular reduce of the partial clerk sums over RCCL at the end of the timed region - the library's own
code behind the C ABI (sda_modular_allreduce_dev; torch.distributed only launches the ranks and carries the
RCCL id).  The headline (config 3) is weak-scaled (100k participants per GPU); attached to it are BASELINE
config 4 (1,000,000 participants of t=2 k=8 n=26) and config 5 (100,000 participants at dim 16,777,216 with
the Lagrange reveal), each with its job sharded over the N ranks.  `rccl` records what carried the exchange;
a communicator that cannot be set up is fatal (exit code 3) unless SDA_SHARE_GPU allows the rehearsal.

`roofline.bound` is the active ceiling ("hbm" / "valu") from profiles/bounds.json (counter passes of this
round; null where no pass was taken); `roofline.frac` is always the HBM fraction; `roofline.kernel` is what the
LIBRARY says it launched (sda_debug_last_kernel), not a guess made here.

Output.  stdout carries ONE COMPACT JSON line (rank 0), at most 4096 bytes - compact_line() below, asserted here and in
the tests: the contract's keys, `roofline`, `cpu_baseline`, and one {value, frac, bound, verified} record per
attached workload.  Everything else (notes, sweeps, per-launch minima and maxima, the full record of every attached
workload) goes to `bench_details.json` beside this script (--details PATH) and to stderr.  Round 4's line had grown to
35 KB and the driver could no longer parse it.  See DESIGN.md "Measurement" for the byte accounting.
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
# the host driver only supports dmabuf IPC; must be in the environment before the HIP runtime starts (RCCL across processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

P62 = 4611686006577364993
OMEGA = {8: 631229665360524489, 9: 3451275676410824977, 16: 2589100645267092065, 27: 365137883145458390,
         256: 3916993753559330817, 729: 4527470848155349462}          # 5^((p-1)/order) mod p
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
    "packed_pss728": dict(kind="packed", n=728, k=100, t=155, o2=256, o3=729, participants=10_000, tile_max=500,
                          desc="tss's shipped shape PSS_155_728_100 (k=100, t=155, n=728; transform kernel), dim 1048576, 62-bit prime"),
    # the reference's own valid domain (tss multiplies i64 residues without widening: p < 2^31.5): the tss-valid shapes over a
    # 31-bit prime (largest prime = 1 mod 432 below 2^31) and tss's shipped PSS_155_728_100 over ITS prime 746497 with ITS roots
    "narrow_ref": dict(kind="packed", n=8, k=3, t=4, participants=100_000, prime=2147482801, w2=495332030, w3=1761729792,
                       desc="reference-valid tss shape t=4 k=3 n=8 over a 31-bit prime (narrow kernels), dim 1048576"),
    "narrow26_ref": dict(kind="packed", n=26, k=8, t=7, participants=100_000, prime=2147482801, w2=1541819067, w3=638656353,
                         desc="reference-valid tss shape t=7 k=8 n=26 over a 31-bit prime (narrow kernels), dim 1048576"),
    "narrow_pss728": dict(kind="packed", n=728, k=100, t=155, participants=10_000, tile_max=500, prime=746497, w2=95660, w3=610121,
                          desc="tss's shipped PSS_155_728_100 (k=100, t=155, n=728) over tss's own prime 746497 and roots: the "
                               "limb GEMM on the matrix cores (ngemm_kernels.hip), dim 1048576"),
    # tss's other shipped parameter set: 19682 clerks, 197 share values per secret (two share buffers of 66 GB at 40 participants)
    "narrow_pss19682": dict(kind="packed", n=19682, k=100, t=155, participants=1_000, tile_max=40, prime=5038849, w2=4318906, w3=1814687,
                            desc="tss's shipped PSS_155_19682_100 (k=100, t=155, n=19682) over tss's own prime 5038849 and roots: the "
                                 "limb GEMM on the matrix cores, dim 1048576"),
    # config 5 on ONE GPU: its 100k participants are spread over 8 GPUs (12.5k each); the dimension is what differs, and
    # the reveal over 16 Mi secrets is part of it.  --dim defaults to 16777216 for this workload.
    "packed_dim16m": dict(kind="packed", n=8, k=3, t=1, o2=8, o3=9, participants=12_500, dim=1 << 24, tile_max=125,
                          desc="BASELINE config 5 per-GPU share: packed Shamir t=1 k=3 n=8, dim 16777216, 62-bit prime, Lagrange reveal"),
}
TILE_MAX = 2500      # participants resident per launch (secrets 21 GB + two share buffers of 56 GB at config 3)


def prime_of(w):
    return w.get("prime", P62)


def roots_of(w):
    """(omega_secrets, omega_shares) of a packed workload"""
    return (w["w2"], w["w3"]) if "w2" in w else (OMEGA[w["o2"]], OMEGA[w["o3"]])


def plan_steps(target_participants, steps, tile_max):
    """A step = one pass of the hot path over one batch of participants.  The batch is target / steps participants
    (rounded up), issued as n_sub resident sub-tiles of p_sub <= tile_max participants each."""
    per_step = max(1, -(-target_participants // steps))
    n_sub = -(-per_step // tile_max)
    for m in range(n_sub, 2 * n_sub + 1):                   # prefer sub-tiles that cover the batch exactly
        if per_step % m == 0:
            return m, per_step // m
    p_sub = -(-per_step // n_sub)
    return n_sub, p_sub


def describe(w, dim, participants_total, world):
    """the label is DERIVED from what the run processes, never a constant"""
    if w["kind"] == "packed":
        shape = f"packed Shamir t={w['t']} k={w['k']} n={w['n']}"
    else:
        shape = f"additive {w['n']}-way"
    bits = prime_of(w).bit_length()
    return f"{shape}, dim {dim}, {bits}-bit prime modulus, {participants_total} participants ({participants_total // world} per GPU)"


def algorithmic_bytes_per_element(n, k):
    """share-gen reads 8 B (secret) and writes 8n/k B; clerk-sum reads 8n/k B."""
    return 8.0 + 8.0 * n / k, 8.0 * n / k


def _host_cpu():
    """model name, physical cores and hardware threads of this box ("stating the core count and CPU model")"""
    model, cores, threads = None, set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            key, _, val = ln.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                threads += 1
            elif key == "model name" and model is None:
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not ln.strip():
                if core is not None:
                    cores.add((phys, core))
                phys = core = None
        if core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"cpu_model": model, "physical_cores": len(cores) or None, "hardware_threads": threads or os.cpu_count(),
            "usable_threads": usable}


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


CPU_SWEEP_CACHE = os.path.join(ROOT, "profiles", "cpu_baseline_hosts.json")


def cpu_baseline(w, dim, budget_s=10.0, samples=3, name="", mode="auto"):
    """The oracle's reference-faithful scalar port (share-gen + clerk-sum) on a bounded sample of the same workload:
    (i) ONE thread like the reference (it has no threading), median of `samples` runs - ALWAYS measured on this box; (ii) the
    same port over participants on the host's cores for a sweep of thread counts, median of `samples` runs each, the BEST
    reported as `all_cores`.  The sweep costs ~100 s of a 256-thread box (two thirds of the default run's wall time in round
    5), so with mode "auto" it is taken from profiles/cpu_baseline_hosts.json when that file holds a sweep of THIS workload
    on THIS host type (same CPU model and thread count; `cached: true` and where it was measured are on the record) and
    measured only on a host the file does not know; mode "fresh" always measures.  Reported, never the thing shipped."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import coracle
    packed = 1 if w["kind"] == "packed" else 0
    a = (packed, prime_of(w), w["n"], w["k"], w["t"], *roots_of(w))
    host = _host_cpu()
    t0 = time.perf_counter()
    coracle.baseline_pass(*a, 1, dim, 0, SEED, KEY)
    one = max(time.perf_counter() - t0, 1e-3)
    parts = max(1, min(1024, int(budget_s / samples / one)))
    rates, secs = [], 0.0
    for i in range(samples):
        t0 = time.perf_counter()
        done, _ = coracle.baseline_pass(*a, parts, dim, i * parts, SEED, KEY)
        dt = time.perf_counter() - t0
        rates.append(done / dt)
        secs += dt
    res = {"value": _median(rates), "unit": "elements/s", "cores": 1, "kind": "port",
           "sample": f"median of {samples} runs of {parts} participants x dim {dim} (share-gen incl. buffered ChaCha20 draws + "
                     f"clerk-sum), oracle/sda_oracle.c single thread, {secs:.1f} s in all",
           "samples": rates,
           "port_notes": "scalar port of the reference's loops in the MATRIX form of packed Shamir (one n x (k+t) "
                         "modular mat-vec per batch, no per-batch allocation) with BUFFERED ChaCha20 draws - both are "
                         "concessions in the reference's favour: tss 0.2 runs two recursive FFTs with a Vec per level "
                         "and the reference makes one OsRng call per draw",
           **host}
    # the same port over participants on many host cores (the reference itself has no threading);
    # each thread owns a participant range and its own clerk sums - the final n x B modular merge is negligible.
    # Oversubscribing the box made round 2's figure worse than round 1's, so sweep and keep the best.
    usable = host["usable_threads"]
    key = f"{host['cpu_model']}|{usable} threads|{name}|dim {dim}"
    if mode == "auto":
        try:
            hit = json.load(open(CPU_SWEEP_CACHE)).get(key)
        except (OSError, ValueError):
            hit = None
        if hit:
            res["all_cores"] = dict(hit, cached=True, cache=f"profiles/cpu_baseline_hosts.json[{key}] (--cpu-baseline fresh re-measures)")
            return res
    counts = sorted({c for c in (64, host["physical_cores"] or 0, usable) if 1 < c <= usable})
    sweep = []
    for threads in counts:
        per = max(1, int(0.25 * budget_s / samples / one))          # ~1 s of work per thread and sample
        rs = []
        for i in range(samples):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                outs = list(ex.map(lambda j: coracle.baseline_pass(*a, per, dim, (i * threads + j) * per, SEED, KEY)[0],
                                   range(threads)))
            rs.append(sum(outs) / (time.perf_counter() - t0))
        sweep.append({"threads": threads, "value": _median(rs), "samples": rs,
                      "sample": f"median of {samples} runs of {threads} threads x {per} participants x dim {dim}"})
    if sweep:
        best = max(sweep, key=lambda e: e["value"])
        res["all_cores"] = {"value": best["value"], "unit": "elements/s", "cores": best["threads"], "sample": best["sample"],
                            "sweep": sweep, "cached": False, "cache_key": key}
    return res


# ---- exit codes of bench.py (DESIGN.md 6 lists them) -------------------------------------------------------------------------
EXIT_OK = 0
EXIT_COMM = 3          # a device is missing or the RCCL communicator could not be set up (and no rehearsal switch allows a fall-back)
EXIT_DEADLINE = 4      # --deadline-s passed: the launcher (or a rank's own watchdog) ended every rank; NO JSON line was printed
EXIT_EXCHANGE = 5      # a cross-rank exchange (communicator set-up or the modular reduce) did not return within its limit
_T0 = time.monotonic()


class Watchdog:
    """Hang protection of ONE rank (the launcher has its own deadline in self_launch): a daemon thread that ends the process
    with a distinct exit code - and says why on stderr - when the whole run passes --deadline-s or the phase in progress passes
    its own limit.  The first multi-GPU run of the exchange must not be able to sit in a collective until the driver's
    timeout: a rank that is ended here exits non-zero, and both launchers (self_launch, torch.distributed.run) then take
    the other ranks down.  phase() also prints the one-line stderr heartbeat (rank, seconds since start, phase).

    Test hook (tests only): SDA_BENCH_TEST_HANG="<phase prefix>:<rank or *>" makes that rank sleep forever when it enters the
    phase - the stand-in for a rank that never comes back from a collective."""

    def __init__(self, rank, world, deadline_s, exchange_timeout_s):
        import threading
        self.rank, self.world = rank, world
        # a rank started by self_launch counts from the LAUNCHER's start and gives it 5 s to act first
        base = float(os.environ.get("SDA_BENCH_LAUNCHER_T0", "nan"))
        self.t0 = _T0
        self.deadline = None
        if deadline_s > 0:
            self.deadline = (_T0 + deadline_s) if base != base else (base + deadline_s + 5.0)
        self.deadline_s = deadline_s
        self.exchange_timeout_s = exchange_timeout_s
        self.name, self.limit_at, self.limit_s, self.code = "start", None, None, EXIT_DEADLINE
        self.diag = lambda: ""                       # Env fills this in: peers, RCCL version, last library error
        self.lock = threading.Lock()
        hang = os.environ.get("SDA_BENCH_TEST_HANG", "")
        self.hang_phase, _, who = hang.partition(":")
        self.hang_me = bool(hang) and who in ("*", str(rank))
        threading.Thread(target=self._watch, name="bench-watchdog", daemon=True).start()

    def phase(self, name, limit_s=None, code=EXIT_DEADLINE):
        with self.lock:
            self.name, self.limit_s, self.code = name, limit_s, code
            self.limit_at = None if not limit_s or limit_s <= 0 else time.monotonic() + limit_s
        print(f"[bench] rank {self.rank}/{self.world} +{time.monotonic() - self.t0:.1f}s phase: {name}"
              + (f" (limit {limit_s:g} s)" if limit_s else ""), file=sys.stderr, flush=True)
        if self.hang_me and name.startswith(self.hang_phase):
            print(f"[bench] rank {self.rank}: SDA_BENCH_TEST_HANG: sleeping forever in phase {name!r}", file=sys.stderr, flush=True)
            while True:
                time.sleep(3600)

    def exchange(self, name):
        """phase with the exchange limit and the exchange exit code"""
        self.phase(name, self.exchange_timeout_s, EXIT_EXCHANGE)

    def _watch(self):
        while True:
            time.sleep(0.25)
            t = time.monotonic()
            with self.lock:
                name, limit_at, limit_s, code = self.name, self.limit_at, self.limit_s, self.code
            why = None
            if limit_at is not None and t > limit_at:
                why = f"phase {name!r} did not finish within its limit of {limit_s:g} s"
            elif self.deadline is not None and t > self.deadline:
                why, code = f"--deadline-s {self.deadline_s:g} passed in phase {name!r}", EXIT_DEADLINE
            if why:
                try:
                    extra = self.diag()
                except Exception as e:                                   # the diagnosis must never keep the process alive
                    extra = f"(diagnosis failed: {e})"
                print(f"[bench] rank {self.rank}/{self.world} WATCHDOG: {why}; {extra}; exiting with code {code}, no JSON line",
                      file=sys.stderr, flush=True)
                os._exit(code)


class Env:
    """process-wide state: device, ranks, library, communicator.

    torch.distributed is used ONLY to launch ranks and as a side channel (gloo: the 128-byte RCCL id, barriers, the
    max-over-ranks of the timing); the data-path reduce is the library's own RCCL code behind the C ABI
    (sda_comm_init / sda_modular_allreduce_dev)."""

    def __init__(self, wd):
        self.wd = wd
        wd.phase("init: import torch, select the device, gloo control plane", 600)
        import torch
        import torch.distributed as dist
        from sda_amd import capi
        self.torch, self.dist, self.capi = torch, dist, capi
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # Device selection and what may replace RCCL:
        #   (default)          rank r runs on device LOCAL_RANK and the cross-GPU reduce is the library's RCCL code.  If the
        #                      communicator cannot be set up the run is FATAL (exit code 3): a number printed over a silent
        #                      host-staged exchange would not be the measurement the line claims.
        #   SDA_SHARE_GPU=1    rehearsal of the N > 1 path on a ONE-GPU box: every rank runs on device 0 and the exchange is
        #                      staged through host memory over gloo (sda_amd.distributed; RCCL refuses two ranks per device);
        #                      the modular sum of the slices still runs on the device.
        #   SDA_SHARE_GPU=try  ranks share device 0, the RCCL communicator is still attempted, its refusal is reported and
        #                      the labelled host-staged exchange takes over.
        #   SDA_BENCH_DEVICE=d pins every rank to device d WITHOUT allowing the fall-back (test of the fatal path).
        share = os.environ.get("SDA_SHARE_GPU", "")
        self.share_gpu = share == "1"
        fallback_allowed = share in ("1", "try")
        if "SDA_BENCH_DEVICE" in os.environ:
            device_index = int(os.environ["SDA_BENCH_DEVICE"])
        else:
            device_index = 0 if share in ("1", "try") else self.local_rank
        if device_index >= torch.cuda.device_count():
            # --gpus N on a box with fewer devices: nothing this run could print would be the N-GPU measurement
            print(f"[bench] rank {self.rank}: device {device_index} does not exist ({torch.cuda.device_count()} visible); FATAL - "
                  f"set SDA_SHARE_GPU=1 only to rehearse the N > 1 path on a one-GPU box", file=sys.stderr, flush=True)
            os._exit(EXIT_COMM)
        torch.cuda.set_device(device_index)
        self.dev = torch.device("cuda", device_index)
        self.use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # launched by torch.distributed.run
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")                                      # control plane only
        # A/B runs (tools/*.sh) and the one-rank RCCL test select non-default kernels by environment variable.  The RELEASE library
        # has no knob table and reads no environment variable: only when such a variable is set does the bench - a measurement
        # tool - load libsda_hip_test.so (same objects + the test hooks, include/sda_hip_debug.h) and hand the values over.  The
        # default line never does: `library` on the line says which binary ran.
        knob_names = ("SDA_FORCE_GENERIC", "SDA_FORCE_MONT64", "SDA_FORCE_FFT", "SDA_FORCE_MFMA", "SDA_NO_MFMA", "SDA_NO_SIDE_STREAM",
                      "SDA_SIDE_STREAM_WGS", "SDA_FFT_G", "SDA_FFT_THREADS", "SDA_FORCE_COLLECTIVES", "SDA_NO_NARROW", "SDA_NO_LAZY", "SDA_NO_NGEMM",
                      "SDA_NO_WIDE_GROUP", "SDA_SIDE_STREAM_PRIORITY", "SDA_NGEMM_CLERK_WG", "SDA_NO_KARATSUBA")
        knobs = {n: os.environ[n] for n in knob_names if os.environ.get(n)}
        if knobs:
            capi.use_test_hooks()
        self.lib = capi.load()
        for name, v in knobs.items():
            capi.check(self.lib.sda_debug_set_knob(name.encode(), int(v) if v.lstrip("-").isdigit() else 1))
        self.csprng_share_map = os.environ.get("SDA_BENCH_SHARE_MAP", "")   # "tss": A/B against the round-3 share map
        capi.check(self.lib.sda_set_device(device_index))
        self.comm = C.c_void_p()
        self.exchange = "none (one rank)" if self.world == 1 else "host-staged over gloo (ranks share one GPU)"
        path = "none (one rank)" if self.world == 1 else "host-staged gloo (SDA_SHARE_GPU rehearsal)"
        self.rccl = {"ranks": 0, "path": path}
        wd.diag = self.diagnosis
        if not self.share_gpu:
            if self.world > 1:
                wd.exchange("comm: RCCL unique id + ncclCommInitRank over %d ranks" % self.world)
            # RCCL prints its version banner to the C stdout at init: keep stdout for the ONE JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            ok = 1
            try:
                ident = (C.c_uint8 * 128)()
                if self.rank == 0 and self.lib.sda_comm_unique_id(ident) != capi.OK:
                    ok = 0
                box = [bytes(ident), ok]
                if self.use_dist:
                    dist.broadcast_object_list(box, src=0)
                ok = box[1]                                          # no id (RCCL not loadable on rank 0): nobody enters the collective init
                if ok and self.lib.sda_comm_init((C.c_uint8 * 128)(*box[0]), self.rank, self.world, C.byref(self.comm)) != capi.OK:
                    ok = 0
                C.CDLL(None).fflush(None)
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            if self.use_dist:                                        # every rank takes the same path
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                if self.world > 1:
                    self.exchange = "RCCL send/recv reduce-scatter + modular-sum kernel + all-gather, behind the C ABI"
                    path = "send/recv"
                elif os.environ.get("SDA_FORCE_COLLECTIVES"):
                    path = "send/recv (one rank, to itself)"
            else:
                msg = self.lib.sda_last_error().decode()
                capi.LAST_ERROR_SEEN = msg
                if self.comm:
                    self.lib.sda_comm_free(self.comm)
                self.comm = C.c_void_p()
                if not (fallback_allowed and self.use_dist):
                    # one device per rank and no communicator: nothing this run could print would be the multi-GPU measurement
                    print(f"[bench] rank {self.rank}: sda_comm_init failed ({msg}); FATAL - set SDA_SHARE_GPU=1 only to "
                          f"rehearse the N > 1 path on a one-GPU box", file=sys.stderr, flush=True)
                    sys.stderr.flush()
                    os._exit(EXIT_COMM)
                print(f"[bench] rank {self.rank}: sda_comm_init failed ({msg}); exchange falls back to host staging over gloo",
                      file=sys.stderr, flush=True)
                self.exchange = "host-staged over gloo (RCCL communicator unavailable: " + msg[:80] + ")"
                path = "host-staged gloo (RCCL refused: " + msg[:60] + ")"
        # machine-readable record of what carried the exchange: ranks the library's communicator spans (0 = none) and the
        # number of distinct physical GPUs (host name + PCI bus id) under the ranks
        import socket
        bus = C.create_string_buffer(64)
        capi.check(self.lib.sda_device_pci_bus_id(device_index, bus, 64))
        mine = (socket.gethostname(), bus.value.decode(), device_index)
        everyone = [None] * self.world
        if self.use_dist:
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        self.rccl = {"ranks": int(self.lib.sda_comm_world(self.comm)) if self.comm else 0,
                     "unique_devices": len({(h, b) for h, b, _ in everyone}),
                     "path": path,
                     "devices": [f"{b} (ordinal {o})" for _, b, o in everyone],
                     "comm_device": int(self.lib.sda_comm_device(self.comm)) if self.comm else None,
                     "rccl_version": int(self.lib.sda_comm_rccl_version())}
        wd.phase("init done: exchange = " + self.exchange)

    def diagnosis(self):
        """what the watchdog prints when it ends this rank: who the peers are, what carries the exchange, the RCCL version the
        library bound and the library's last error on the main thread (as of the last call that failed)"""
        r = self.rccl
        return (f"device {self.dev}, peers {r.get('devices', '?')}, exchange path {r.get('path')!r}, communicator ranks "
                f"{r.get('ranks')}, RCCL version {int(self.lib.sda_comm_rccl_version())}, "
                f"last library error {self.capi.LAST_ERROR_SEEN!r}")

    def modular_allreduce(self, t, q=P62):
        """sum over ranks mod q of the int64 device tensor `t`, on every rank (new tensor)"""
        torch = self.torch
        if self.use_dist and not self.comm:
            from sda_amd.distributed import modular_allreduce
            return modular_allreduce(t, q)
        out = torch.empty_like(t)
        self.capi.check(self.lib.sda_modular_allreduce_dev(self.comm, q, t.data_ptr(), t.numel(), out.data_ptr(),
                                                           torch.cuda.current_stream(self.dev).cuda_stream or None))
        return out

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, x: float) -> float:
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.torch.cuda.synchronize(self.dev)
        if self.use_dist:
            self.dist.barrier()
        if self.comm:
            self.lib.sda_comm_free(self.comm)
        if self.use_dist:
            self.dist.destroy_process_group()


def _setup(env, name, dim, P, row_align, rounds):
    from sda_amd import crypto
    w = WORKLOADS[name]
    n, k, t = w["n"], w["k"], w["t"]
    scheme = (crypto.PackedShamir(k, n, t, prime_of(w), *roots_of(w)) if w["kind"] == "packed"
              else crypto.Additive(n, prime_of(w)))
    B = (dim + k - 1) // k
    Bs = (B + row_align - 1) // row_align * row_align      # 128-byte aligned rows (row_align = 16 elements)
    gen = crypto.ShareGenerator(scheme)
    gen.set_drbg_key(KEY)                                   # deterministic mode: reproducible run, verified below
    if env.csprng_share_map == "tss" and w["kind"] == "packed":
        gen.set_csprng_share_map(gen.SHARE_MAP_TSS_NODES)
    if rounds != 20:
        gen.set_drbg_rounds(rounds)
    comb = crypto.ShareCombiner(scheme)
    return w, n, k, t, scheme, B, Bs, gen, comb


def _share_map_name(gen, w):
    if w["kind"] != "packed":
        return "n/a (additive: shares 0..n-2 are the draws, additive.rs:42-47)"
    return ("systematic (library): the t draws of a batch are its shares 0..t-1, n-t dot products per batch"
            if gen.csprng_share_map() == gen.SHARE_MAP_SYSTEMATIC else
            "tss nodes (reference): draws = values at omega_secrets^(k+1..k+t), n dot products per batch")


def _library_kernel(env):
    """what the library launched in its last generate call on this thread (include/sda_hip_debug.h) - the bench never guesses"""
    return env.lib.sda_debug_last_kernel().decode()


def _expected_sums(env, secrets, P, dim, firsts, q=P62):
    """column sums mod p of the secrets the run shared, summed over the ranks: `firsts` = the first participant index of
    every tile THIS rank processed.  One entry repeated K times is the replayed resident tile (no refill needed); distinct
    entries regenerate each tile with the bench's own fill kernel (splitmix64 of (participant, component))."""
    torch, capi, lib, dev = env.torch, env.capi, env.lib, env.dev
    from sda_amd import crypto
    cs = crypto.ShareCombiner(crypto.Additive(2, q))
    cs.begin_dev(1, dim)
    resident = None
    for first in firsts:
        if first != resident:
            capi.check(lib.sda_fill_synthetic_dev(secrets.data_ptr(), P, dim, dim, first, SEED, q, None))
            resident = first
        cs.update_dev(secrets.data_ptr(), 0, P, dim)
    exp = torch.empty(dim, dtype=torch.int64, device=dev)
    cs.finish_dev(exp.data_ptr())
    return env.modular_allreduce(exp, q)


def _verify(env, scheme, secrets, total, P, dim, B, firsts, q=P62):
    """size-independent check of the full result: reconstruct(clerk sums over all ranks) == the column sums of every
    secret vector that was shared, mod p.  Also times the reveal (Lagrange reconstruction over `dim` secrets,
    receive.rs:140-152) with HIP events."""
    import ctypes as C
    torch, capi, lib, dev = env.torch, env.capi, env.lib, env.dev
    from sda_amd import crypto
    rec = crypto.SecretReconstructor(scheme, dim)
    out = torch.empty(dim, dtype=torch.int64, device=dev)
    idx = list(range(scheme.reconstruction_threshold()))
    rows = total[:len(idx)].contiguous()
    rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)      # builds the Lagrange matrix once
    torch.cuda.synchronize(dev)
    e0, e1 = C.c_void_p(), C.c_void_p()
    capi.check(lib.sda_event_create(C.byref(e0))); capi.check(lib.sda_event_create(C.byref(e1)))
    reps = 5
    capi.check(lib.sda_event_record(e0, None))
    for _ in range(reps):
        rec.reconstruct_dev(idx, rows.data_ptr(), B, B, out.data_ptr(), dim)
    capi.check(lib.sda_event_record(e1, None))
    ms = C.c_float()
    capi.check(lib.sda_event_elapsed_ms(e0, e1, C.byref(ms)))
    lib.sda_event_destroy(e0); lib.sda_event_destroy(e1)
    reveal_ms = ms.value / reps
    exp_total = _expected_sums(env, secrets, P, dim, firsts, q)
    torch.cuda.synchronize(dev)
    verified = bool(torch.equal(out, exp_total))
    nrows = len(idx)
    reveal = {"ms": reveal_ms, "secrets_per_s": dim / (reveal_ms * 1e-3), "dim": dim, "clerk_rows": nrows,
              "algorithmic_bytes": 8 * (nrows * B + dim),
              "GBps": 8 * (nrows * B + dim) / (reveal_ms * 1e-3) / 1e9,
              "note": "Lagrange reconstruction of the dim secrets from t+k clerk sums (receive.rs:140-152): average of 5 "
                      "reconstruct_dev calls between HIP events, outside the timed region"}
    return verified, reveal


def _line(env, name, w, dim, n, k, t, P, n_sub, steps, warmup, dt, Bs, rounds, schedule, share_map=None):
    world = env.world
    participants_total = world * steps * n_sub * P
    elements = float(participants_total) * dim
    value = elements / dt
    gen_b, comb_b = algorithmic_bytes_per_element(n, k)
    return {
        "metric": "share-gen + clerk-sum elements/sec (mod q)", "value": value, "unit": "elements/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak",
        "scaling_note": "WEAK: every GPU processes the whole per-GPU workload named in config.workload (BASELINE config 3: 100,000 "
                        "participants PER GPU unless --participants says otherwise), so `value` grows with N at fixed time; the "
                        "legs under additional_workloads at N > 1 (BASELINE configs 4 and 5) are STRONG-scaled instead: their "
                        "job is fixed and its participants are divided over the N ranks (each leg says so in its own `scaling`)",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": describe(w, dim, participants_total, world), "baseline_config": w["desc"], "name": name,
                   "dim": dim, "participants_total": participants_total, "participants_per_step_per_gpu": n_sub * P,
                   "sub_tiles_per_step": n_sub, "tile_participants": P,
                   "share_count": n, "secret_count": k, "privacy_threshold": t, "modulus": prime_of(w),
                   "randomness": f"on-device ChaCha{rounds} (sda-drbg-v1, deterministic bench key)",
                   "csprng_share_map": share_map,
                   "row_stride_elements": Bs, "schedule": schedule,
                   "parallelism": f"participants sharded x{world}, one modular reduce of the clerk sums at the end",
                   "exchange": env.exchange},
        "path_roofline": {"bytes_per_element": gen_b + comb_b,
                          "achieved_GBps": value / world * (gen_b + comb_b) / 1e9,
                          "frac_of_hbm_peak": value / world * (gen_b + comb_b) / 1e9 / HBM_PEAK_GBS},
    }


def _profiles_json(fname):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", fname)))
    except Exception:
        return {}


def _traffic(name, P, dim, key):
    """PMC HBM bytes per launch - a MEASUREMENT of exactly this (workload, tile, dimension) or None.  profiles/traffic.json is
    written by tools/make_traffic.py from the round's rocprofv3 --pmc passes (FETCH_SIZE doubled as the microarchitecture
    guide prescribes, WRITE_SIZE as reported), one entry per profiled (workload, tile, dim); a run at any other tile gets
    `traffic: null` rather than a scaled figure (round 3 scaled; a changed default tile would then have extrapolated silently)."""
    v = _profiles_json("traffic.json").get(f"{name}:tile{P}:dim{dim}")
    if not v:
        return None
    if key in v:
        return v[key]
    if key == "fused_bytes_per_launch" and "gen_bytes_per_launch" in v and "comb_bytes_per_launch" in v:
        return v["gen_bytes_per_launch"] + v["comb_bytes_per_launch"]                      # two launches per call
    return None


def _traffic_note(name, P, dim, got):
    if got is not None:
        return f"measured: profiles/traffic.json[{name}:tile{P}:dim{dim}] (rocprofv3 --pmc passes of this command form)"
    have = sorted(k for k in _profiles_json("traffic.json") if k.startswith(name + ":"))
    return (f"null: no counter pass was taken at tile {P}, dim {dim} for this workload "
            f"(profiled: {', '.join(have) if have else 'none'}); not extrapolated")


# which ceiling is active when no counter evidence has been collected for a workload (profiles/bounds.json missing)
BOUND_WITHOUT_EVIDENCE = {"packed": "hbm", "packed_dim16m": "hbm"}


def _bound(name, role, roof):
    """SURVEY.md 8d "report which bound is active": the verdict of tools/make_bounds.py for this workload and launch form
    (profiles/bounds.json, from the round's rocprofv3 SQ / FETCH / WRITE counter passes): "hbm" only where the measured HBM
    traffic / duration is within 10 % of the floor tools/microbench_hbm reaches with the same access pattern and no
    arithmetic; otherwise "valu", with the VALU wave-instructions per element, the SIMD cycles available per issued VALU
    instruction and the VALU-busy fraction the counters give.  `frac` stays the HBM fraction either way."""
    entry = _profiles_json("bounds.json").get(name, {})
    # a shape without a dual-role kernel runs share-gen and clerk-sum as two launches (side by side on two streams): the
    # share-gen kernel is the dominant one and its counters are the evidence
    e = entry.get(role) or (entry.get("serial_gen") if role == "fused" else None)
    if not e:
        return {"bound": BOUND_WITHOUT_EVIDENCE.get(name), "bound_evidence": None}    # no counter pass: no claim
    out = {"bound": e["bound"], "bound_evidence": e.get("evidence")}
    if "valu" in e:
        out["valu"] = e["valu"]
    if "hbm" in e:
        out["hbm"] = e["hbm"]
    return out


# (<= 120 characters each: they travel on the compact stdout line)
INPUT_MODES = {
    # one resident tile of synthetic participants per GPU, generated on the device before timing, re-shared by every sub-tile
    # with fresh share randomness (inputs resident in HBM when the timed region starts)
    "replay": "replay: one resident tile per GPU (filled on device before timing), re-shared by every sub-tile, fresh randomness",
    # every sub-tile shares DIFFERENT participants (splitmix64 of (participant, component)): two secret buffers,
    # tile i+1 generated on a side stream INSIDE the timed region while tile i runs (+8 B written per element, not counted)
    "distinct": "distinct: every sub-tile shares different participants, next tile filled on a side stream inside the timed region",
}


def measure_fused(env, name, dim, P, n_sub, steps, warmup, row_align=16, verify=True, rounds=20, inputs="replay"):
    """Software-pipelined schedule: ONE dual-role launch per sub-tile generates sub-tile i while the clerk sums of
    sub-tile i-1 are accumulated (sda_share_generator_generate_combine_dev); K tiles take K + 1 launches.
    A step = n_sub sub-tiles of P participants."""
    torch, capi, lib, dev = env.torch, env.capi, env.lib, env.dev
    rank, world = env.rank, env.world
    w, n, k, t, scheme, B, Bs, gen, comb = _setup(env, name, dim, P, row_align, rounds)
    q = prime_of(w)
    distinct = inputs == "distinct"
    # ONE allocation for the resident working set (secrets, the two share buffers), the buffers 1 MiB apart inside it.  With three
    # separate allocations the physical placement of the two share buffers relative to each other is a lottery per process: in
    # about half the runs every second launch (the ones that write the second buffer) took 23.1 ms instead of 22.0 - 114 - 116
    # Gelem/s where 118.5 - 119.5 is the kernel's rate (profiles/r06/headline_launch_pattern.txt, headline_arena.txt).
    # SDA_BENCH_ARENA_PAD=<bytes> sets another gap, -1 the separate allocations (experiments only).
    arena_pad = int(os.environ.get("SDA_BENCH_ARENA_PAD", str(1 << 20)))
    n_sec = 2 if distinct else 1
    if arena_pad >= 0:
        e_sec, e_sh, gap = P * dim, n * P * Bs, arena_pad // 8
        arena = torch.empty(n_sec * (e_sec + gap) + 2 * (e_sh + gap) + 64, dtype=torch.int64, device=dev)
        secrets = [arena[j * (e_sec + gap):j * (e_sec + gap) + e_sec].view(P, dim) for j in range(n_sec)]
        off = n_sec * (e_sec + gap)
        shares = [arena[off + i * (e_sh + gap):off + i * (e_sh + gap) + e_sh].view(n, P, Bs) for i in range(2)]
    else:
        arena = None
        secrets = [torch.empty((P, dim), dtype=torch.int64, device=dev) for _ in range(n_sec)]
        shares = [torch.empty((n, P, Bs), dtype=torch.int64, device=dev) for _ in range(2)]
    side = torch.cuda.Stream(dev) if distinct else None
    filled = [torch.cuda.Event() for _ in secrets]          # secrets[j] holds its tile
    consumed = [torch.cuda.Event() for _ in secrets]        # the launch that read secrets[j] has finished

    def first_of(i):
        """first participant index of tile i on this rank (= its CSPRNG stream base)"""
        return (i * world + rank) * P

    def fill(i, stream=None):
        buf = secrets[i % len(secrets)]
        src = first_of(i) if distinct else rank * P
        capi.check(lib.sda_fill_synthetic_dev(buf.data_ptr(), P, dim, dim, src, SEED, q, stream))

    kern_seen = [None]

    def launch(i, total, ev=None):
        """launch i of total+1: generate tile i (if i < total), sum tile i-1 (if i > 0)"""
        cur, prev = shares[i % 2], shares[(i - 1) % 2]
        j = i % len(secrets)
        if distinct and i < total:
            torch.cuda.current_stream(dev).wait_event(filled[j])
        if ev:
            capi.check(lib.sda_event_record(ev[0], None))
        gen.generate_combine_dev(comb, secrets[j].data_ptr(), P if i < total else 0, dim, dim, cur.data_ptr(), Bs, P * Bs,
                                 d_prev=prev.data_ptr() if i > 0 else 0, prev_participants=P if i > 0 else 0,
                                 first_participant=first_of(i))
        if ev:
            capi.check(lib.sda_event_record(ev[1], None))
        if 0 < i < total and kern_seen[0] is None:
            kern_seen[0] = _library_kernel(env)
        if distinct and i + 1 < total:
            # tile i+1 goes into the OTHER buffer, last read by launch i-1 (already ordered before this point on the
            # launch stream): generate it on the side stream while launch i runs
            consumed[j].record(torch.cuda.current_stream(dev))
            jn = (i + 1) % 2
            side.wait_event(consumed[jn])
            with torch.cuda.stream(side):
                fill(i + 1, side.cuda_stream)
                filled[jn].record(side)

    def prime(total):
        fill(0)
        for e in filled + consumed:
            e.record(torch.cuda.current_stream(dev))

    wd = env.wd
    wd.phase(f"warm-up: {name}, {warmup} step(s) of {n_sub} x {P} participants, dim {dim}")
    wtiles = warmup * n_sub
    comb.begin_dev(n, B)
    prime(wtiles)
    for i in range(wtiles + 1):
        launch(i, wtiles)
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B)                                     # discard the warm-up contributions
    tiles = steps * n_sub
    prime(tiles)                                             # tile 0 resident before the timed region starts
    evs = []
    for _ in range(2 * (tiles + 1)):
        e = C.c_void_p()
        capi.check(lib.sda_event_create(C.byref(e)))
        evs.append(e)
    sums = torch.zeros((n, B), dtype=torch.int64, device=dev)
    if world > 1:
        # the FIRST exchange between the devices (nothing before this line has moved a byte from one GPU to another): under the
        # exchange watchdog, so that a deadlocked send/recv group ends the run with EXIT_EXCHANGE and a diagnosis in seconds
        wd.exchange(f"warm-up exchange: {name}, modular reduce of {8 * n * B} bytes per GPU over {world} ranks")
    env.modular_allreduce(sums, q)                           # connect RCCL outside the timed region, real message size
    torch.cuda.synchronize(dev)
    env.barrier()
    wd.phase(f"timed: {name}, {steps} step(s) of {n_sub} x {P} participants")
    t0 = time.perf_counter()
    for i in range(tiles + 1):
        launch(i, tiles, evs[2 * i:2 * i + 2])
    comb.finish_dev(sums.data_ptr())
    torch.cuda.synchronize(dev)
    if world > 1:
        wd.exchange(f"timed exchange: {name}")
    tx = time.perf_counter()
    total = env.modular_allreduce(sums, q)                   # X1: the only exchange step
    torch.cuda.synchronize(dev)
    exchange_ms = (time.perf_counter() - tx) * 1e3
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    wd.phase(f"verify: {name}")
    ms = C.c_float()
    launch_ms = []
    for i in range(tiles + 1):
        capi.check(lib.sda_event_elapsed_ms(evs[2 * i], evs[2 * i + 1], C.byref(ms)))
        launch_ms.append(ms.value)
    all_ms = sum(launch_ms) / len(launch_ms)                 # what rocprofv3 --stats averages (K+1 launches)
    full = launch_ms[1:tiles]                                # launches that carry both roles
    full_ms = sum(full) / len(full) if full else None        # needs >= 2 tiles
    for e in evs:
        lib.sda_event_destroy(e)
    firsts = [first_of(i) for i in range(tiles)] if distinct else [rank * P] * tiles
    verified, reveal = _verify(env, scheme, secrets[0], total, P, dim, B, firsts, q) if verify else (None, None)
    gen_b, comb_b = algorithmic_bytes_per_element(n, k)
    per_launch_bytes = P * dim * (gen_b + comb_b)
    gbs = tiles * per_launch_bytes / (sum(launch_ms) * 1e-3) / 1e9
    kern = kern_seen[0] or _library_kernel(env)              # a launch that carried both roles, as the library names it
    has_dual = " + " not in kern
    res = _line(env, name, w, dim, n, k, t, P, n_sub, steps, warmup, dt, Bs, rounds,
                "dual-role launch: share-gen of tile i and clerk-sum of tile i-1 interleaved in one grid (shares "
                "materialised in HBM by one launch, read back by the next); K+1 launches for K tiles" if has_dual else
                "sda_share_generator_generate_combine_dev without a dual-role kernel for this shape: clerk-sum of tile i-1, then "
                "share-gen of tile i, two launches per call", share_map=_share_map_name(gen, w))
    res["config"]["inputs"] = INPUT_MODES[inputs]
    res["config"]["distinct_participants"] = world * tiles * P if distinct else world * P
    res["config"]["library_path"] = gen.path_name()
    res["roofline"] = {"kernel": kern, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": gbs / HBM_PEAK_GBS, "traffic": _traffic(name, P, dim, "fused_bytes_per_launch"),
                       "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": all_ms,
                       "launches": tiles + 1, "both_roles_launch_ms": full_ms,
                       # SURVEY.md 8d asks for a median over >= 5 repetitions: `value` is total elements / wall time (the
                       # contract), these are the per-launch HIP-event durations behind it
                       "both_roles_launch_ms_median": _median(full) if full else None,
                       "both_roles_launch_ms_min_max": [min(full), max(full)] if full else None,
                       "first_launch_ms_share_gen_only": launch_ms[0], "last_launch_ms_clerk_sum_only": launch_ms[-1],
                       "launch_ms_each": [round(x, 3) for x in launch_ms],          # (details file only: every launch, in order)
                       "note": "one launch = share-gen of a tile (8 + 8n/k B/element) + clerk-sum of the previous tile "
                               "(8n/k B/element); K tiles take K+1 launches (the first only generates, the last only "
                               "sums), achieved = K x algorithmic_bytes_per_launch / sum of the K+1 launch durations"}
    res["roofline"]["traffic_note"] = _traffic_note(name, P, dim, res["roofline"]["traffic"])
    res["roofline"].update(_bound(name, "fused", res["roofline"]) if rounds == 20 else
                           {"bound": None, "bound_evidence": f"no counter pass was taken at ChaCha{rounds} (A/B leg; the product runs ChaCha20)"})
    res["verified_reconstruct_equals_sum"] = verified
    res["verified_against"] = ("column sums of all %d distinct participants" % (world * tiles * P) if distinct else
                               "%d x (column sums of the %d resident participants of every rank)" % (tiles, P)) if verify else None
    res["reveal"] = reveal
    res["exchange_ms"] = env.max_over_ranks(exchange_ms)      # the cross-GPU modular reduce, inside the timed region
    res["exchange_bytes_per_gpu"] = 8 * n * B
    res["config"]["allocation"] = ("one arena, buffers %d bytes apart" % arena_pad) if arena_pad >= 0 else "separate allocations"
    del secrets, shares, sums, total, arena
    torch.cuda.empty_cache()
    return res


def measure(env, name, dim, P, n_sub, steps, warmup, row_align=16, overlap=0, verify=True, rounds=20):
    """serial schedule: share-gen launch then clerk-sum launch per sub-tile (clean per-kernel timings)"""
    torch, capi, lib, dev = env.torch, env.capi, env.lib, env.dev
    rank, world = env.rank, env.world
    w, n, k, t, scheme, B, Bs, gen, comb = _setup(env, name, dim, P, row_align, rounds)
    q = prime_of(w)
    if overlap:
        comb.set_residency(2)            # 8 waves per CU saturate HBM; the rest stay with share generation
    nbuf = 2 if overlap else 1
    secrets = torch.empty((P, dim), dtype=torch.int64, device=dev)
    shares = [torch.empty((n, P, Bs), dtype=torch.int64, device=dev) for _ in range(nbuf)]
    if overlap:
        s_gen, s_comb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    else:
        s_gen = s_comb = torch.cuda.current_stream(dev)
    h_gen, h_comb = s_gen.cuda_stream or None, s_comb.cuda_stream or None
    capi.check(lib.sda_fill_synthetic_dev(secrets.data_ptr(), P, dim, dim, rank * P, SEED, q, None))
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B, h_comb or 0)
    gen_done = [torch.cuda.Event() for _ in range(nbuf)]
    comb_done = [torch.cuda.Event() for _ in range(nbuf)]

    def step(i, slot, evs=None):
        first = (i * world + rank) * P
        buf = shares[slot]
        if overlap:
            s_gen.wait_event(comb_done[slot])
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
    env.wd.phase(f"warm-up: {name} (serial schedule)")
    tiles = steps * n_sub
    for i in range(warmup * n_sub):
        step(tiles + i, i % nbuf)
    torch.cuda.synchronize(dev)
    comb.begin_dev(n, B, h_comb or 0)
    for e in comb_done:
        e.record(s_comb)
    evs = []
    for _ in range(4 * tiles):
        e = C.c_void_p()
        capi.check(lib.sda_event_create(C.byref(e)))
        evs.append(e)
    sums = torch.zeros((n, B), dtype=torch.int64, device=dev)
    wd = env.wd
    if world > 1:
        wd.exchange(f"warm-up exchange: {name}, modular reduce of {8 * n * B} bytes per GPU over {world} ranks")
    env.modular_allreduce(sums, q)
    torch.cuda.synchronize(dev)
    env.barrier()
    wd.phase(f"timed: {name} (serial schedule), {steps} step(s) of {n_sub} x {P} participants")
    t0 = time.perf_counter()
    for i in range(tiles):
        step(i, i % nbuf, evs[4 * i:4 * i + 4])
    comb.finish_dev(sums.data_ptr(), h_comb or 0)
    torch.cuda.synchronize(dev)
    if world > 1:
        wd.exchange(f"timed exchange: {name}")
    tx = time.perf_counter()
    total = env.modular_allreduce(sums, q)
    torch.cuda.synchronize(dev)
    exchange_ms = (time.perf_counter() - tx) * 1e3
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    wd.phase(f"verify: {name}")
    gen_ms = comb_ms = 0.0
    ms = C.c_float()
    for i in range(tiles):
        capi.check(lib.sda_event_elapsed_ms(evs[4 * i], evs[4 * i + 1], C.byref(ms))); gen_ms += ms.value
        capi.check(lib.sda_event_elapsed_ms(evs[4 * i + 2], evs[4 * i + 3], C.byref(ms))); comb_ms += ms.value
    gen_ms /= tiles
    comb_ms /= tiles
    for e in evs:
        lib.sda_event_destroy(e)
    verified, reveal = _verify(env, scheme, secrets, total, P, dim, B, [rank * P] * tiles, q) if verify else (None, None)
    gen_b, comb_b = algorithmic_bytes_per_element(n, k)
    per_launch = P * dim
    gen_gbs = per_launch * gen_b / (gen_ms * 1e-3) / 1e9
    comb_gbs = per_launch * comb_b / (comb_ms * 1e-3) / 1e9
    dominant_gen = gen_ms >= comb_ms
    gen_kernel = _library_kernel(env)                        # the last generate_batch_dev call, as the library names it
    res = _line(env, name, w, dim, n, k, t, P, n_sub, steps, warmup, dt, Bs, rounds,
                "share-gen(i+1) overlapped with clerk-sum(i) on two streams, double-buffered shares" if overlap
                else "one stream, serial", share_map=_share_map_name(gen, w))
    res["config"]["inputs"] = INPUT_MODES["replay"]
    res["config"]["library_path"] = gen.path_name()
    res["roofline"] = {"kernel": gen_kernel if dominant_gen else "combine_update_kernel",
                       "achieved": gen_gbs if dominant_gen else comb_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": (gen_gbs if dominant_gen else comb_gbs) / HBM_PEAK_GBS,
                       "traffic": _traffic(name, P, dim, "gen_bytes_per_launch" if dominant_gen else "comb_bytes_per_launch"),
                       "algorithmic_bytes_per_launch": per_launch * (gen_b if dominant_gen else comb_b),
                       "avg_launch_ms": gen_ms if dominant_gen else comb_ms}
    res["roofline"]["traffic_note"] = _traffic_note(name, P, dim, res["roofline"]["traffic"])
    res["roofline"].update(_bound(name, "serial_gen" if dominant_gen else "serial_comb", res["roofline"]))
    res["kernels"] = {"share_gen": {"avg_ms": gen_ms, "bytes_per_element": gen_b, "GBps": gen_gbs,
                                    "frac_of_hbm_peak": gen_gbs / HBM_PEAK_GBS},
                      "clerk_sum": {"avg_ms": comb_ms, "bytes_per_element": comb_b, "GBps": comb_gbs,
                                    "frac_of_hbm_peak": comb_gbs / HBM_PEAK_GBS}}
    res["verified_reconstruct_equals_sum"] = verified
    res["reveal"] = reveal
    res["exchange_ms"] = env.max_over_ranks(exchange_ms)      # the cross-GPU modular reduce, inside the timed region
    res["exchange_bytes_per_gpu"] = 8 * n * B
    del secrets, shares, sums, total
    torch.cuda.empty_cache()
    return res


LINE_LIMIT = 4096        # bytes of the ONE stdout line (the driver stopped parsing somewhere above 32 KB; stay far below)


def _short(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _sig(x, digits=6):
    """floats to `digits` significant figures (the full-precision values are in the details file)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _pick(d, keys, digits=6, keep_null=("traffic", "bound")):
    return {k: _sig(_short(d[k]), digits) for k in keys if k in d and (d[k] is not None or k in keep_null)}


def compact_line(full, details_path="bench_details.json"):
    """The ONE stdout line: the contract's keys + roofline + cpu_baseline + a four-field record per attached workload; strings
    cut to 120 characters, floats to 6 significant figures.  Everything `full` holds beyond that is in the details file."""
    line = {k: _sig(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data") if k in full}
    line["config"] = _pick(full.get("config", {}), ("workload", "baseline_config", "name", "dim", "participants_total",
                                                     "tile_participants", "modulus", "csprng_share_map", "inputs", "library_path"))
    roof = full.get("roofline") or {}
    line["roofline"] = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "tables_current",
                                    "algorithmic_bytes_per_launch", "both_roles_launch_ms", "avg_launch_ms", "launches"))
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "cpu_model", "physical_cores"))
        if cpu.get("all_cores"):
            line["cpu_baseline"]["all_cores"] = _pick(cpu["all_cores"], ("value", "cores", "cached"))
    else:
        line["cpu_baseline"] = None
    line["verified_reconstruct_equals_sum"] = full.get("verified_reconstruct_equals_sum")
    if full.get("n_gpus", 1) > 1 and full.get("rccl"):
        line["rccl"] = _pick(full["rccl"], ("ranks", "unique_devices", "path"))
        line["exchange_ms"] = _sig(full.get("exchange_ms"), 4)
    extra = full.get("additional_workloads") or {}
    if extra:
        line["additional_workloads"] = {
            name: {"value": _sig(r.get("value"), 5), "frac": _sig(r.get("frac_wall", (r.get("roofline") or {}).get("frac")), 4),
                   "bound": (r.get("roofline") or {}).get("bound"), "verified": r.get("verified_reconstruct_equals_sum"),
                   **({"reveal_ms": _sig(r["reveal"]["ms"], 4)} if name.startswith("config5") and r.get("reveal") else {}),
                   **({"rounds": r["rounds"]} if "rounds" in r else {}),
                   **({"frac_with_fill": _sig(r["frac_with_fill"], 4)} if "frac_with_fill" in r else {})}
            for name, r in extra.items()}
    line["build_id"] = full.get("build_id")
    line["details"] = os.path.basename(details_path)
    return line


def emit(full, details_path, fd, full_line=False):
    """details file + stderr first, then the compact line on `fd` (the real stdout), its size checked"""
    try:
        with open(details_path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
    except OSError as e:                                     # a read-only tree must not cost the measurement
        print(f"[bench] could not write {details_path}: {e}", file=sys.stderr)
    print("[bench] full record:", json.dumps(full), file=sys.stderr, flush=True)
    if full_line:                                            # tools/*.sh only: the whole record as the stdout line
        os.write(fd, (json.dumps(full) + "\n").encode())
        return
    text = json.dumps(compact_line(full, details_path), separators=(",", ":"))
    if len(text.encode()) > LINE_LIMIT:
        raise SystemExit(f"bench.py: the stdout line is {len(text.encode())} bytes (> {LINE_LIMIT}); trim compact_line()")
    os.write(fd, (text + "\n").encode())


def self_launch(n, deadline_s=0.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks here - the same script, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them, stdout inherited (only rank 0 writes to it:
    the ONE JSON line) - wait for all of them and return the exit code of the rank that failed FIRST (so a refused
    communicator stays EXIT_COMM, a stuck exchange EXIT_EXCHANGE).  A rank that dies takes the others with it after a grace
    period instead of leaving them in a collective; and when `deadline_s` passes with ranks still running - a hang nobody
    exited from - every rank is ended (SIGTERM, then SIGKILL) and the launcher returns EXIT_DEADLINE.  Rank 0 prints its line
    only as its very last act, so a run that ends this way has printed none."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                SDA_BENCH_LAUNCHER_T0=repr(_T0),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")) for r in range(n)]

    def stop(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()
    signal.signal(signal.SIGTERM, stop)
    def kill_all(grace=5.0):
        stop()
        t_end = time.monotonic() + grace
        while any(p.poll() is None for p in procs) and time.monotonic() < t_end:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
    first_bad, failed_at, stopped = 0, None, False
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            if deadline_s > 0 and time.monotonic() - _T0 > deadline_s:
                alive = [r for r, p in enumerate(procs) if p.poll() is None]
                print(f"[bench] launcher: --deadline-s {deadline_s:g} passed with rank(s) {alive} still running; ending all "
                      f"{n} ranks, exit code {EXIT_DEADLINE}, no JSON line (each rank's last '[bench] rank r/N ... phase:' "
                      f"line on stderr says where it was)", file=sys.stderr, flush=True)
                kill_all()
                for p in procs:
                    p.wait()
                return first_bad or EXIT_DEADLINE
            if not first_bad:
                bad = [c for c in (p.poll() for p in procs) if c not in (None, 0)]
                if bad:
                    first_bad = 128 - bad[0] if bad[0] < 0 else bad[0]   # the rank that failed FIRST is the cause
                    failed_at = time.monotonic()
            if failed_at is not None and not stopped and time.monotonic() - failed_at > 15.0:
                stop()                                   # the survivors wait for a rank that will never come
                stopped = True
    except KeyboardInterrupt:
        stop()
        raise
    codes = [p.wait() for p in procs]
    if first_bad:
        return first_bad
    return max([128 - c if c < 0 else c for c in codes] + [0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="packed", choices=sorted(WORKLOADS))
    ap.add_argument("--dim", type=int, default=0, help="vector dimension (default: the workload's, 1048576 unless stated)")
    ap.add_argument("--participants", type=int, default=0,
                    help="participants per GPU over the whole run (default: the BASELINE configuration's count); a step "
                         "covers participants / steps of them, issued as resident sub-tiles of at most --tile")
    ap.add_argument("--tile", type=int, default=0, help=f"largest resident sub-tile, participants (default {TILE_MAX})")
    ap.add_argument("--row-align", type=int, default=16, help="pad share rows to a multiple of this many elements")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: share-gen of tile i+1 runs concurrently with clerk-sum of tile i (two streams, "
                         "double-buffered shares, clerk-sum capped at 2 workgroups per CU); 0: one stream, serial")
    ap.add_argument("--schedule", default="fused", choices=["serial", "fused"],
                    help="serial: share-gen launch then clerk-sum launch per tile; fused: one dual-role launch per "
                         "tile (tile i generated while tile i-1 is summed)")
    ap.add_argument("--drbg-rounds", type=int, default=20, choices=[20, 12, 8],
                    help="ChaCha rounds of the on-device CSPRNG (A/B only; the product runs ChaCha20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "fresh"],
                    help="the ONE-thread CPU baseline is always timed on this box; the all-cores thread sweep (~100 s) is taken from "
                         "profiles/cpu_baseline_hosts.json when it knows this host type and workload (auto) or always measured (fresh)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-additional", action="store_true",
                    help="skip the attached runs (N = 1: short config-2 and config-5-shape runs; N > 1: BASELINE configs 4 and 5 "
                         "sharded over the ranks)")
    ap.add_argument("--inputs", default="replay", choices=sorted(INPUT_MODES),
                    help="replay: one resident tile re-shared by every sub-tile (inputs resident before timing); distinct: "
                         "every sub-tile shares different participants, generated on a side stream inside the timed region")
    ap.add_argument("--leg-participants", type=int, default=0,
                    help="config 4 / config 5 legs (N = 1: full job on one GPU; N > 1: job sharded over the ranks): participants "
                         "of the WHOLE job (default: the BASELINE configuration's: 1,000,000 for config 4, 100,000 for "
                         "config 5); rehearsals pass something small")
    ap.add_argument("--leg-dim", type=int, default=0, help="config 4 / config 5 legs: vector dimension (default: the configuration's)")
    ap.add_argument("--full-line", action="store_true",
                    help="measurement scripts only (tools/*.sh): print the FULL record on stdout instead of the compact line")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"),
                    help="where the full record goes (stdout carries the compact line only)")
    ap.add_argument("--deadline-s", type=float, default=900.0,
                    help=f"hang protection: end every rank and exit with code {EXIT_DEADLINE} (no JSON line) when the whole run "
                         "takes longer than this many seconds; 0 = no deadline (long custom runs)")
    ap.add_argument("--exchange-timeout-s", type=float, default=180.0,
                    help=f"hang protection: limit of ONE cross-rank exchange (communicator set-up, the warm-up and the timed "
                         f"modular reduce); a rank that waits longer prints its diagnosis and exits with code {EXIT_EXCHANGE}")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: become the launcher of its own N ranks (one process per GPU); the ranks
        # are this same script with the environment torch.distributed.run would give them
        raise SystemExit(self_launch(args.gpus, args.deadline_s))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: one rank per GPU")
    # stdout carries ONE JSON line and nothing else: gloo and RCCL print banners to the C stdout of every rank, so the
    # process's fd 1 points at stderr for the whole run and the line is written to the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    wd = Watchdog(int(os.environ.get("RANK", "0")), world, args.deadline_s, args.exchange_timeout_s)
    wd.phase("start")
    env = Env(wd)

    def run(name, steps, warmup, participants=0, dim=0, tile=0, rounds=0, inputs=None):
        w = WORKLOADS[name]
        dim = dim or w.get("dim", 1 << 20)
        n_sub, p_sub = plan_steps(participants or w["participants"], steps, tile or w.get("tile_max", TILE_MAX))
        if args.schedule == "fused" and not args.overlap:
            return measure_fused(env, name, dim, p_sub, n_sub, steps, warmup, args.row_align, verify=not args.no_verify,
                                 rounds=rounds or args.drbg_rounds, inputs=inputs or args.inputs)
        return measure(env, name, dim, p_sub, n_sub, steps, warmup, args.row_align, args.overlap, verify=not args.no_verify,
                       rounds=rounds or args.drbg_rounds)

    line = run(args.workload, args.steps, args.warmup, args.participants, args.dim, args.tile)
    line["rccl"] = env.rccl
    line["build_id"] = env.lib.sda_build_id().decode()
    line["kernel_id"] = env.lib.sda_kernel_id().decode()
    # `roofline.traffic` / `roofline.bound` are look-ups in profiles/{traffic,bounds}.json: say whether those tables were measured
    # on the kernels that are running now (their `_kernel_id` stamp against sda_kernel_id()) - a kernel change that forgot to
    # regenerate them would otherwise print a stale figure
    stamps = {f: _profiles_json(f).get("_kernel_id") for f in ("traffic.json", "bounds.json")}
    line["roofline"]["tables_current"] = all(v == line["kernel_id"] for v in stamps.values())
    line["roofline"]["tables_kernel_id"] = stamps
    line["library"] = env.lib.sda_version().decode()
    keep = ("value", "unit", "n_gpus", "steps", "ms_per_step", "config", "kernels", "roofline", "path_roofline",
            "verified_reconstruct_equals_sum", "verified_against", "reveal", "exchange_ms", "exchange_bytes_per_gpu")
    if env.world == 1 and not args.no_additional and args.workload == "packed":
        # BASELINE config 2 (additive 3-way, 10k participants = 5 steps of 2000)
        add = run("additive", 5, 2)
        # BASELINE configs 4 and 5 at their FULL job size on this one GPU (they are defined on 8 GPUs; with one GPU they are
        # "the largest single-GPU configuration" and fit by streaming resident tiles): config 4 = 1,000,000 participants of
        # t=2 k=8 n=26 as 800 tiles of 1250 (~15 s); config 5 = 100,000 participants at dim 16,777,216 as 800 tiles of 125,
        # Lagrange reveal over the 16 Mi secrets included (~16 s).  --leg-participants / --leg-dim shrink them (tests).
        job4 = args.leg_participants or 1_000_000
        job5 = args.leg_participants or 100_000
        c4 = run("packed26", 10, 1, participants=job4, dim=args.leg_dim or 0, tile=1250)
        c4["config"]["job"] = f"BASELINE config 4 at full job size on ONE GPU: {job4} participants, streamed as resident tiles"
        big = run("packed_dim16m", 10, 1, participants=job5, dim=args.leg_dim or 0, tile=125)
        big["config"]["job"] = (f"BASELINE config 5 at full job size on ONE GPU: {job5} participants at dim "
                                f"{big['config']['dim']}, Lagrange reveal included (reveal.ms)")
        # tss's own shipped parameter set PSS_155_728_100 (k=100 t=155 n=728) through the transform kernel, clerk sum on the
        # side stream (4 steps of 500 participants)
        pss = run("packed_pss728", 4, 1, participants=2000)
        line["additional_workloads"] = {"additive": {k: add[k] for k in keep if k in add},
                                        "config4_full": {k: c4[k] for k in keep if k in c4},
                                        "config5_full": {k: big[k] for k in keep if k in big},
                                        "packed_pss728": {k: pss[k] for k in keep if k in pss}}
        # the reference's OWN valid domain (tss multiplies i64 residues without widening): the tss-valid shapes over a 31-bit
        # prime and tss's shipped PSS_155_728_100 over its own prime 746497, through the narrow (one 32-bit limb) kernels
        # (tiles of 2000 for the two small shapes: 2 - 4 % above tiles of 1500, profiles/r06/narrow_tile_sweep.txt.  TWELVE tiles per
        # leg: K tiles take K + 1 launches - the first only generates, the last only sums - and with round 5's four tiles that
        # pipeline fill was a fifth of a leg's time, 0.61 on the line for a kernel whose both-roles launches run at 0.65)
        for nm, part, tile in (("narrow_ref", 24000, 2000), ("narrow26_ref", 24000, 2000), ("narrow_pss728", 6000, 500),
                               ("narrow_pss19682", 480, 40)):
            r = run(nm, 12, 1, participants=part, tile=tile)
            line["additional_workloads"][nm] = {k: r[k] for k in keep if k in r}
        # the REFERENCE's own share map on the headline shape (packed_shamir.rs:42 -> tss share: values = [0] ++ secrets ++
        # randomness): every other number on this line is on the library's systematic map (n - t dot products per batch)
        if not env.csprng_share_map:
            env.csprng_share_map = "tss"
            r = run("packed", 10, 1, participants=(args.participants or 25_000), dim=args.dim)
            env.csprng_share_map = ""
            line["additional_workloads"]["packed_tss_nodes"] = {k: r[k] for k in keep if k in r}
        # The price of 100,000 DIFFERENT participants (participate.rs:37-76 runs once per participant; the headline replays one
        # resident tile): every sub-tile shares other participants, the next tile's secrets are generated on a side stream INSIDE
        # the timed region.  That fill writes 8 B per element which the path's 50.67 B/element do not count: `frac` is on the
        # path's bytes (comparable with the headline), `frac_with_fill` counts the fill's bytes as well.
        if args.inputs != "distinct":
            r = run("packed", 5, 1, participants=(args.participants or 12_500), dim=args.dim, inputs="distinct")
            gb, cb = algorithmic_bytes_per_element(r["config"]["share_count"], r["config"]["secret_count"])
            r["fill_bytes_per_element"] = 8.0
            # WALL-clock fractions for this leg: its launches run at the headline's speed, what it pays is the time they wait for
            # the fill (roofline.frac, from the launch durations, would hide exactly that)
            r["frac_wall"] = r["path_roofline"]["frac_of_hbm_peak"]
            r["frac_with_fill"] = r["frac_wall"] * (gb + cb + 8.0) / (gb + cb)
            line["additional_workloads"]["packed_distinct"] = {k: r[k] for k in keep + ("fill_bytes_per_element", "frac_wall", "frac_with_fill") if k in r}
        # The price of the round count, labelled: configs 2 and 4 are bound by the vector ALUs on ChaCha20 (two / 0.25 64-bit draws
        # per element); the same legs with the CSPRNG at 12 rounds show how much of their gap to the headline is that policy.  The
        # PRODUCT runs 20 rounds (what chacha.rs:36 - rand 0.3's ChaChaRng - uses) and these two are never the headline.
        if args.drbg_rounds == 20:
            r = run("additive", 5, 2, rounds=12)                                 # config 2's own job: 10,000 participants, 5 tiles
            r["rounds"] = 12
            line["additional_workloads"]["additive_chacha12"] = {k: r[k] for k in keep + ("rounds",) if k in r}
            r = run("packed26", 20, 1, participants=(args.leg_participants or 50_000), dim=args.leg_dim or 0, tile=1250, rounds=12)
            r["rounds"] = 12
            line["additional_workloads"]["config4_chacha12"] = {k: r[k] for k in keep + ("rounds",) if k in r}
    if env.world > 1 and not args.no_additional and args.workload == "packed":
        # The two BASELINE configurations that are DEFINED on several GPUs, sharded over the ranks that
        # are here: config 4 = 1,000,000 participants of packed Shamir t=2 k=8 n=26; config 5 = 100,000 participants at
        # dim 16,777,216 with the Lagrange reveal.  Whole job / world participants per GPU (strong scaling in N for these
        # two: the job is fixed), each rank's clerk sums meeting in ONE modular reduce over RCCL inside the timed region
        # (combiner.rs:20-26 is what is being sharded).  `value` = all participants x dim / max-over-ranks time.
        legs = {}
        for leg, tile, leg_steps in (("packed26", 1500, 10), ("packed_dim16m", 125, 10)):
            w = WORKLOADS[leg]
            job = args.leg_participants or (1_000_000 if leg == "packed26" else 100_000)
            per_gpu = -(-job // env.world)
            r = run(leg, leg_steps, 1, participants=per_gpu, dim=args.leg_dim or 0, tile=tile)
            r["scaling"] = "strong (the job's participants / world per GPU)"
            r["config"]["job"] = (f"BASELINE config {4 if leg == 'packed26' else 5}: {job} participants over {env.world} GPUs "
                                  f"= {per_gpu} per GPU" + (", Lagrange reveal included (reveal.ms)" if leg == "packed_dim16m" else ""))
            legs["config4_packed26" if leg == "packed26" else "config5_packed_dim16m"] = {k: r[k] for k in keep + ("scaling",) if k in r}
        line["additional_workloads"] = legs
    wd.phase("cpu baseline on rank 0's host cores" if env.rank == 0 and not args.no_cpu_baseline else "waiting for rank 0")
    if env.rank == 0:
        # rank 0's host cores, at any world size (the other ranks wait at the closing barrier): the reference's CPU path
        # timed beside the GPU figure on the same box
        line["cpu_baseline"] = (None if args.no_cpu_baseline else
                                cpu_baseline(WORKLOADS[args.workload], args.dim or WORKLOADS[args.workload].get("dim", 1 << 20),
                                             name=args.workload, mode=args.cpu_baseline))
    if env.use_dist:
        env.barrier()                                        # (the other ranks wait here, under the run's deadline only)
    # tear the communicator down with the C stdout pointed at stderr (RCCL may print there), so that the JSON line is
    # the one and LAST thing on stdout
    wd.phase("close: barrier, communicator teardown", 120 if env.world > 1 else None, EXIT_EXCHANGE)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        env.close()
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    wd.phase("emit")
    if env.rank == 0:
        emit(line, args.details, real_stdout, args.full_line)
    os.close(real_stdout)


if __name__ == "__main__":
    main()
