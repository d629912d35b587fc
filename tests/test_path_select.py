"""Kernel selection as a TABLE (no GPU): sda_amd/csrc/path_select.hpp is the one place where the library decides which
kernel family serves a packed-Shamir scheme; sda_debug_select_path() runs that function on the host.  The reference
dispatches on the scheme enum alone (client/src/crypto/sharing/mod.rs:37-53), so every row below is a performance
decision between bit-exact kernels - the table is its documentation, and a new kernel family is one new row.

Also here: the build id baked into libsda_hip.so equals the digest of the sources in the tree (a stale prebuilt library
cannot pass for a fresh one), and bench.py's ONE stdout line stays below 4096 bytes whatever is attached to it."""
import ctypes as C
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P62 = 4611686006577364993
OMEGA = {8: 631229665360524489, 9: 3451275676410824977, 16: 2589100645267092065, 27: 365137883145458390,
         256: 3916993753559330817, 729: 4527470848155349462}          # 5^((p-1)/order) mod p (bench.py)
P31 = 2147482801                                                       # largest prime = 1 mod 432 below 2^31
TSS_P, TSS_W2, TSS_W3 = 746497, 95660, 610121                          # tss's shipped PSS_155_728_100
TSS_P2, TSS2_W2, TSS2_W3 = 5038849, 4318906, 1814687                   # tss's shipped PSS_155_19682_100


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from sda_amd import capi
    return capi.hooks_library()          # sda_debug_select_path (stateless) lives in libsda_hip_test.so only


def select(lib, k, t, n, p, w2, w3, knobs=""):
    from sda_amd import capi
    s = capi.SharingScheme(capi.SHARING_PACKED_SHAMIR, n, p, k, t, w2, w3)
    buf = C.create_string_buffer(512)
    st = lib.sda_debug_select_path(C.byref(s), knobs.encode(), buf, 512)
    assert st == capi.OK, lib.sda_last_error().decode()
    return dict(kv.split("=") for kv in buf.value.decode().split())


# (k, t, n, p, omega_secrets, omega_shares)  ->  wide family, narrow overlay, limb-31 radix (0 = not limb-31)
TABLE = [
    # BASELINE configs 3 / 5 and 4 over the 62-bit prime: compiled limb-31 kernels ((8,2) in the three-digit form)
    ((3, 1, 8, P62, OMEGA[8], OMEGA[9]), ("l31", "none", 62)),
    ((8, 2, 26, P62, OMEGA[16], OMEGA[27]), ("l31", "none", 93)),
    # their tss-valid neighbours: three-digit limb-31
    ((3, 4, 8, P62, OMEGA[8], OMEGA[9]), ("l31", "none", 93)),
    ((8, 7, 26, P62, OMEGA[16], OMEGA[27]), ("l31", "none", 93)),
    # another compiled tss-valid split of k + t = 7
    ((4, 3, 8, P62, OMEGA[8], OMEGA[9]), ("l31", "none", 62)),
    # 12..16 terms without a three-digit instance: the 62-bit limb GEMM on the matrix cores
    ((12, 3, 26, P62, OMEGA[16], OMEGA[27]), ("mfma", "none", 0)),
    # tss's PSS_155_728_100 over a wide prime: tss's own transform
    ((100, 155, 728, P62, OMEGA[256], OMEGA[729]), ("fft", "none", 0)),
    # p < 2^31 and k + t <= 16 (everything the reference itself can run): one-limb kernels on top of the wide family
    ((3, 4, 8, P31, 495332030, 1761729792), ("l31", "n31", 93)),
    ((8, 7, 26, P31, 1541819067, 638656353), ("l31", "n31", 93)),
    ((3, 4, 8, 433, 354, 150), ("l31", "n31", 93)),                      # full_loop.rs:54-67
    # p <= 0x7F7F7F and k + t > 16: tss's two shipped parameter sets run as the narrow limb GEMM
    ((100, 155, 728, TSS_P, TSS_W2, TSS_W3), ("fft", "ngemm", 0)),
    ((100, 155, 19682, TSS_P2, TSS2_W2, TSS2_W3), ("fft", "ngemm", 0)),
    # no transform structure, k + t <= 64: limb-31 with the matrix in global memory; beyond: the any-shape kernel
    ((20, 13, 80, P62, 3, 5), ("l31_global", "none", 0)),
    ((40, 30, 100, P62, 3, 5), ("generic", "none", 0)),
]


@pytest.mark.parametrize("scheme,want", TABLE, ids=[f"k{s[0]}t{s[1]}n{s[2]}p{s[3].bit_length()}" for s, _ in TABLE])
def test_selection_table(lib, scheme, want):
    got = select(lib, *scheme)
    assert (got["wide"], got["narrow"], int(got["r_bits"])) == want, got
    # what each kind of call runs follows from the choice: the narrow overlay serves ChaCha20 and injected randomness,
    # the wide family serves the other round counts (A/B only)
    top = want[1] if want[1] != "none" else want[0]
    assert got["call20"] == top and got["injected"] == top and got["call12"] == want[0]
    dual = {"l31", "mfma", "n31", "ngemm"}
    assert got["fused20"] == (top if top in dual else "none")
    assert got["fused12"] == (want[0] if want[0] in dual else "none")


def test_knobs_move_the_selection_and_nothing_else_does(lib):
    base = (3, 1, 8, P62, OMEGA[8], OMEGA[9])
    assert select(lib, *base, knobs="SDA_FORCE_GENERIC")["wide"] == "generic"
    assert select(lib, *base, knobs="SDA_FORCE_MONT64")["wide"] == "mont64"
    assert select(lib, 3, 4, 8, P62, OMEGA[8], OMEGA[9], knobs="SDA_FORCE_FFT")["wide"] == "fft"
    assert select(lib, 8, 7, 26, P62, OMEGA[16], OMEGA[27], knobs="SDA_FORCE_MFMA")["wide"] == "mfma"
    assert select(lib, 12, 3, 26, P62, OMEGA[16], OMEGA[27], knobs="SDA_NO_MFMA")["wide"] == "l31"
    pss = (100, 155, 728, TSS_P, TSS_W2, TSS_W3)
    assert select(lib, *pss, knobs="SDA_NO_NGEMM")["narrow"] == "none"
    assert select(lib, *pss, knobs="SDA_NO_NARROW") == dict(select(lib, *pss, knobs="SDA_NO_NARROW"), wide="fft", narrow="none")
    assert select(lib, 3, 4, 8, P31, 495332030, 1761729792, knobs="SDA_NO_NARROW")["narrow"] == "none"
    # the process-wide knob state is NOT read by the table function (and a live handle snapshots it at creation)
    lib.sda_debug_set_knob(b"SDA_FORCE_GENERIC", 1)
    try:
        assert select(lib, *base)["wide"] == "l31"
    finally:
        lib.sda_debug_reset_knobs()
    from sda_amd import capi
    s = capi.SharingScheme(capi.SHARING_PACKED_SHAMIR, 8, P62, 3, 1, OMEGA[8], OMEGA[9])
    buf = C.create_string_buffer(64)
    assert lib.sda_debug_select_path(C.byref(s), b"SDA_NOT_A_KNOB", buf, 64) == capi.ERR_INVALID_ARGUMENT
    # roots of the wrong order: no transform structure, so a 255-term shape falls to the any-shape kernel
    assert select(lib, 100, 155, 728, P62, 3, 5)["wide"] == "generic"
    add = capi.SharingScheme(capi.SHARING_ADDITIVE, 3, P62, 0, 0, 0, 0)
    buf = C.create_string_buffer(256)
    assert lib.sda_debug_select_path(C.byref(add), None, buf, 256) == capi.OK and b"wide=additive" in buf.value


def test_build_id_is_the_digest_of_the_tree(lib):
    import __graft_entry__ as ge
    assert lib.sda_build_id().decode() == ge.source_digest()
    assert len(ge.source_digest()) == 16 and ge.source_digest() != "unknown"
    # every source and header the build compiles is in the digest (a file added to the build but not to the lists would
    # make a stale library look fresh)
    csrc = os.path.join(ROOT, "sda_amd", "csrc")
    on_disk = {f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".hpp"))}
    assert on_disk == set(ge.SOURCES) | set(ge.HEADERS), on_disk ^ (set(ge.SOURCES) | set(ge.HEADERS))
    assert set(ge.PUBLIC_HEADERS) == {f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")}


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_line", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fat_record(n_gpus=1):
    """a full record like the ones measure_fused() builds, with EVERY attached workload and long notes everywhere"""
    note = "x" * 900
    leg = {"value": 1.23456789e11, "unit": "elements/s", "n_gpus": n_gpus, "steps": 10, "ms_per_step": 12.3456789,
           "config": {"workload": note, "job": note, "dim": 1 << 24, "share_count": 26, "schedule": note, "inputs": note},
           "roofline": {"kernel": "fused_packed_l31_kernel<8, 2, 20>", "achieved": 4912.123456, "peak": 8000.0, "unit": "GB/s",
                        "frac": 0.61401543, "traffic": None, "traffic_note": note, "bound": "valu", "bound_evidence": note, "note": note,
                        "valu": {"wave_instructions_per_element": 5.34, "busy": 0.951}, "hbm": {"frac_of_floor": 0.8}},
           "path_roofline": {"bytes_per_element": 60.0}, "verified_reconstruct_equals_sum": True, "verified_against": note,
           "reveal": {"ms": 0.0631928, "note": note}, "exchange_ms": 0.33, "exchange_bytes_per_gpu": 61070464}
    names = ["additive", "config4_full", "config5_full", "packed_pss728", "narrow_ref", "narrow26_ref", "narrow_pss728",
             "narrow_pss19682", "packed_tss_nodes", "packed_distinct", "additive_chacha12", "config4_chacha12", "config4_packed26",
             "config5_packed_dim16m", "one_more_leg_with_a_long_name"]
    full = dict(leg, metric="share-gen + clerk-sum elements/sec (mod q)", warmup=5, higher_is_better=True, scaling="weak",
                scaling_note=note, vs_baseline=None, dtype="u64", data="synthetic", build_id="0123456789abcdef",
                additional_workloads={n: dict(json.loads(json.dumps(leg)), **({"rounds": 12} if n.endswith("chacha12") else {}),
                                              **({"frac_with_fill": 0.7123456, "fill_bytes_per_element": 8.0} if n == "packed_distinct" else {}))
                                      for n in names},
                cpu_baseline={"value": 18196000.123, "unit": "elements/s", "cores": 1, "kind": "port", "sample": note, "samples": [1.0] * 9,
                              "port_notes": note, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "physical_cores": 128,
                              "hardware_threads": 256, "usable_threads": 256,
                              "all_cores": {"value": 2.81e8, "cores": 64, "sample": note, "sweep": [{"threads": 64, "samples": [1.0] * 3}] * 3}},
                rccl={"ranks": n_gpus, "unique_devices": n_gpus, "path": "send/recv", "devices": [note] * 8, "comm_device": 0})
    full["config"].update(baseline_config=note, name="packed", participants_total=100000, tile_participants=2500, modulus=P62,
                          csprng_share_map=note, library_path="l31")
    return full


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_bench_stdout_line_stays_small_and_carries_the_contract(n_gpus):
    """BENCH_r04.json came back with parsed: null because the line had grown to 35 KB.  compact_line() is what reaches stdout:
    at most 4096 bytes with every leg attached, strings cut to 120 characters, and still every key the contract names."""
    bench = _bench()
    full = _fat_record(n_gpus)
    assert len(json.dumps(full)) > 30000                                   # the record itself is as fat as round 4's
    line = bench.compact_line(full, "/somewhere/bench_details.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text.encode()) <= bench.LINE_LIMIT == 4096, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"}
    assert line["roofline"]["traffic"] is None and line["roofline"]["bound"] == "valu"
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and line["cpu_baseline"]["all_cores"]["cores"] == 64
    assert line["config"]["workload"].endswith("~") and len(line["config"]["workload"]) == 120
    assert all(set(v) <= {"value", "frac", "bound", "verified", "reveal_ms", "rounds", "frac_with_fill"} for v in line["additional_workloads"].values())
    assert len(line["additional_workloads"]) == 15 and line["details"] == "bench_details.json"
    assert line["additional_workloads"]["config4_chacha12"]["rounds"] == 12 and line["additional_workloads"]["packed_distinct"]["frac_with_fill"] == 0.7123
    assert ("rccl" in line) == (n_gpus > 1)
    assert json.loads(text) == line                                        # strict JSON


def test_bench_emit_refuses_an_oversized_line(tmp_path, capfd):
    bench = _bench()
    full = _fat_record()
    full["additional_workloads"] = {f"leg_{i:03d}_{'n' * 40}": full["additional_workloads"]["additive"] for i in range(60)}
    r, w = os.pipe()
    with pytest.raises(SystemExit) as e:
        bench.emit(full, str(tmp_path / "d.json"), w)
    os.close(w)
    assert os.read(r, 10) == b"" and "4096" in str(e.value)                # nothing reached "stdout"
    os.close(r)
    assert json.load(open(tmp_path / "d.json"))["metric"] == full["metric"]   # the details file was still written


def test_bench_cpu_sweep_cache_is_keyed_by_host_and_workload(tmp_path, monkeypatch):
    """bench.py's all-cores CPU sweep (100 s of round 5's 163 s default run) is taken from profiles/cpu_baseline_hosts.json only for
    the SAME cpu model, thread count, workload and dimension; the one-core baseline is always timed; `fresh` ignores the file"""
    bench = _bench()
    host = bench._host_cpu()
    w, dim = dict(bench.WORKLOADS["additive"]), 4096
    key = f"{host['cpu_model']}|{host['usable_threads']} threads|additive|dim {dim}"
    cache = tmp_path / "hosts.json"
    cache.write_text(json.dumps({key: {"value": 123.0, "unit": "elements/s", "cores": 7, "sample": "from the file"}}))
    monkeypatch.setattr(bench, "CPU_SWEEP_CACHE", str(cache))
    got = bench.cpu_baseline(w, dim, budget_s=0.3, samples=1, name="additive", mode="auto")
    assert got["cores"] == 1 and got["value"] > 0 and got["all_cores"]["cached"] is True and got["all_cores"]["value"] == 123.0
    other = bench.cpu_baseline(w, 2048, budget_s=0.3, samples=1, name="additive", mode="auto")          # another dimension: not in the file
    assert other["value"] > 0 and not (other.get("all_cores") or {}).get("cached", False)
    fresh = bench.cpu_baseline(w, dim, budget_s=0.3, samples=1, name="additive", mode="fresh")
    assert not (fresh.get("all_cores") or {}).get("cached", False)
    committed = json.load(open(os.path.join(ROOT, "profiles", "cpu_baseline_hosts.json")))
    assert any(k.endswith("|packed|dim 1048576") for k in committed if not k.startswith("_"))
