"""N > 1 path on CPU: world_size-2 gloo processes run the participant sharding and the modular
all-reduce choreography (all_to_all of slices -> local modular sum -> all_gather) of
sda_amd/distributed.py.  The local modular sum on a GPU is a HIP kernel; here a checker reducer is
injected (there is no CPU fallback in the product), so what is tested is the exchange itself:
slicing, padding, ordering and the no-overflow property that a plain int64 SUM would violate."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P62 = 4611686006577364993


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _checker_modsum(parts: torch.Tensor, modulus: int) -> torch.Tensor:
    """exact column sum mod q with Python ints (checker only)"""
    a = parts.numpy().astype(object)
    return torch.from_numpy(np.array([int(x) % modulus for x in a.sum(axis=0)], dtype=np.int64))


def _worker(rank, world, port, length, q_out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sda_amd.distributed import modular_allreduce, shard_participants
    from oracle import coracle

    # every rank owns a shard of participants and computes its partial clerk sums with the ORACLE
    # (this test is about the exchange; the GPU tests cover the kernels)
    total_participants, n, k, t, dim = 11, 8, 3, 1, length
    first, count = shard_participants(total_participants, world, rank)
    w2, w3 = 631229665360524489, 3451275676410824977
    key = bytes(range(32))
    B = (dim + k - 1) // k
    partial = np.zeros((n, B), dtype=np.int64)
    for p in range(first, first + count):
        secrets = coracle.fill_synthetic(1, dim, p, 7, P62)[0]
        rnd = coracle.drbg_fill(key, p, B, t, P62)
        shares = coracle.packed_generate(P62, k, t, n, w2, w3, secrets, rnd)
        partial = np.stack([coracle.combine(P62, np.stack([partial[c], shares[c]])) for c in range(n)])
    got = modular_allreduce(torch.from_numpy(partial), P62, local_modsum=_checker_modsum)
    # a worst-case vector too: every rank contributes q-1 everywhere (plain int64 SUM wraps for 8 ranks)
    worst = modular_allreduce(torch.full((5, 7), P62 - 1, dtype=torch.int64), P62, local_modsum=_checker_modsum)
    if rank == 0:
        q_out.put((got.numpy(), worst.numpy(), first, count))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,length", [(2, 50), (2, 7), (3, 20)])
def test_modular_allreduce_and_sharding_gloo(world, length):
    from oracle import coracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, length, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, worst, first, count = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth over all 11 participants
    n, k, t, dim = 8, 3, 1, length
    w2, w3 = 631229665360524489, 3451275676410824977
    key = bytes(range(32))
    B = (dim + k - 1) // k
    allshares = []
    for p in range(11):
        secrets = coracle.fill_synthetic(1, dim, p, 7, P62)[0]
        allshares.append(coracle.packed_generate(P62, k, t, n, w2, w3, secrets, coracle.drbg_fill(key, p, B, t, P62)))
    want = np.stack([coracle.combine(P62, np.stack([s[c] for s in allshares])) for c in range(n)])
    assert np.array_equal(got, want)
    assert np.array_equal(worst, np.full((5, 7), (world * (P62 - 1)) % P62, dtype=np.int64))
    # reconstruct from the reduced sums == sum of all secrets
    rec = coracle.packed_reconstruct(P62, k, t, w2, w3, dim, [0, 2, 5, 7], got[[0, 2, 5, 7]])
    assert np.array_equal(rec, coracle.combine(P62, coracle.fill_synthetic(11, dim, 0, 7, P62)))


def test_shard_participants_partition():
    from sda_amd.distributed import shard_participants
    for total in (0, 1, 7, 100_000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_participants(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_single_rank_requires_gpu_reducer():
    """Without an injected reducer the local sum is the HIP kernel: CPU tensors are refused, loudly."""
    from sda_amd.distributed import modular_allreduce
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        modular_allreduce(torch.zeros(4, dtype=torch.int64), P62)


@pytest.mark.parametrize("world,length", [(1, 10), (2, 7), (3, 20), (8, 1000), (8, 3), (5, 0), (8, 8), (4, 22369)])
def test_comm_plan_multi_process_cpp(world, length, tmp_path):
    """The C++ choreography the library runs over RCCL (sda_amd/csrc/comm_plan.hpp, driven by sda_comm.cpp), with an
    injected socket transport between forked processes and a checker reducer: ragged / empty slices, worst-case
    residues (a u64 sum collective would wrap), in-place gather."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "comm_plan_test.cpp")
    exe = str(tmp_path / "comm_plan_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", src, "-o", exe])
    out = subprocess.run([exe, str(world), str(length), str(P62)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "comm plan ok" in out.stdout


def test_bench_leg_planning_covers_the_baseline_jobs_exactly():
    """bench.py --gpus N: BASELINE config 4 (1,000,000 participants) and config 5 (100,000) sharded over N = 2, 4, 8 ranks -
    every rank's share splits into whole sub-tiles within the resident-tile bound, and steps x sub-tiles x tile x ranks is
    exactly the job (the label is derived from that product); the headline's 100,000 per GPU likewise for the driver's
    and the default step counts"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_planning", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (2, 4, 8):
        for job, tile, steps in ((1_000_000, 1500, 10), (100_000, 125, 10)):
            per_gpu = -(-job // world)
            n_sub, p_sub = bench.plan_steps(per_gpu, steps, tile)
            assert p_sub <= tile and steps * n_sub * p_sub * world == job, (world, job, n_sub, p_sub)
    for steps in (20, 50):
        n_sub, p_sub = bench.plan_steps(100_000, steps, bench.TILE_MAX)
        assert p_sub <= bench.TILE_MAX and steps * n_sub * p_sub == 100_000
    # a job that does not divide is rounded UP (never fewer participants than the label says), still inside the bound
    n_sub, p_sub = bench.plan_steps(1001, 10, 7)
    assert p_sub <= 7 and 10 * n_sub * p_sub >= 1001
    # exchange sizes stated in DESIGN.md 6: [n][B] partial sums of 8 bytes
    assert 8 * 26 * -(-(1 << 20) // 8) == 27_262_976 and 8 * 8 * -(-(1 << 24) // 3) == 357_913_984


def test_bench_starts_its_own_ranks_and_reports_the_first_failure():
    """`python bench.py --gpus 2` with no launcher (the command form the driver uses): the script becomes the launcher of its
    two ranks.  Without a GPU every rank fails at device selection - the launcher must come back promptly with a non-zero
    exit code and NO JSON line (never hang in a rendezvous, never print a partial result).  The GPU suite covers the
    successful run (test_bench_launches_its_own_ranks)."""
    import subprocess
    import sys
    import time
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("covered by the GPU test on a box with a device")
    except ImportError:
        pass
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 240


def _bench_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                             "SDA_SHARE_GPU", "SDA_BENCH_TEST_HANG", "SDA_BENCH_LAUNCHER_T0")}
    env.update(extra)
    return env


def test_bench_launcher_deadline_ends_a_run_in_which_a_rank_never_returns():
    """`python bench.py --gpus 2` in which the ranks never come back (SDA_BENCH_TEST_HANG: every rank sleeps
    forever when it enters its first phase - the stand-in for a send/recv group that deadlocks between two devices).  Nobody
    EXITS, so the first-failure rule never fires: the launcher's --deadline-s must end all ranks and come back with the
    distinct exit code 4 (EXIT_DEADLINE), no JSON line on stdout, and a stderr that says where each rank was."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--deadline-s", "6"], capture_output=True, text=True, timeout=120,
                         env=_bench_env(SDA_BENCH_TEST_HANG="start:*"), cwd=root)
    took = time.time() - t0
    assert out.returncode == 4, (out.returncode, out.stderr[-2000:])
    assert "{" not in out.stdout, out.stdout
    assert took < 6 + 20, took                                   # the deadline, the grace of SIGTERM -> SIGKILL, interpreter start-up
    assert "launcher: --deadline-s 6 passed with rank(s) [0, 1] still running" in out.stderr
    for r in (0, 1):                                             # the heartbeat: every rank's last phase line
        assert f"[bench] rank {r}/2 " in out.stderr and "phase: start" in out.stderr


def test_bench_rank_watchdog_ends_a_hung_rank_without_a_launcher():
    """the same hang with NO launcher of ours around the rank (the driver starts N > 1 runs with torch.distributed.run, which
    only reacts to a rank that exits): the rank's own watchdog thread ends it at --deadline-s with exit code 4 and a
    diagnosis on stderr; torch.distributed.run then takes the other ranks down."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--deadline-s", "4"], capture_output=True, text=True, timeout=120,
                         env=_bench_env(SDA_BENCH_TEST_HANG="start:0"), cwd=root)
    assert out.returncode == 4, (out.returncode, out.stderr[-2000:])
    assert "{" not in out.stdout and time.time() - t0 < 30
    assert "WATCHDOG: --deadline-s 4 passed in phase 'start'" in out.stderr and "exiting with code 4, no JSON line" in out.stderr


def test_bench_phase_limit_has_its_own_exit_code():
    """a phase with a limit of its own (the cross-rank exchanges: communicator set-up, warm-up and timed modular reduce) ends
    the rank with exit code 5 (EXIT_EXCHANGE) when it overruns - checked on the watchdog class itself, in a child process that
    sits in such a phase"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog(1, 8, 600.0, 1.5)\n"
            "wd.diag = lambda: 'peers [a, b], RCCL version 22105'\n"
            "wd.exchange('warm-up exchange: test')\n"
            "time.sleep(60)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=_bench_env(), cwd=root)
    assert out.returncode == 5, (out.returncode, out.stderr)
    assert "rank 1/8 WATCHDOG: phase 'warm-up exchange: test' did not finish within its limit of 1.5 s" in out.stderr
    assert "peers [a, b], RCCL version 22105" in out.stderr and "phase: warm-up exchange: test (limit 1.5 s)" in out.stderr
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_codes", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert (bench.EXIT_COMM, bench.EXIT_DEADLINE, bench.EXIT_EXCHANGE) == (3, 4, 5)
    design = open(os.path.join(root, "DESIGN.md")).read()
    for word in ("EXIT_COMM", "EXIT_DEADLINE", "EXIT_EXCHANGE", "--deadline-s", "--exchange-timeout-s"):
        assert word in design, word + " is not documented in DESIGN.md 6"
