"""The narrow limb GEMM's row loop must not touch scratch memory (not a GPU test: hipcc cross-compiles the kernel to assembly here).

Why a test: a B fragment that the register allocator parks in scratch memory and reloads inside the row loop costs more than the
reload - the s_waitcnt vmcnt(0) in front of its use also waits for every share store the wave has in flight, and the kernel has a
dedicated loader wave precisely so that its compute waves never wait for vmcnt there.  Builds that differed only in unrelated code
ran 124 k or 180 k cycles per workgroup in the row-tile phase (DESIGN.md 4 "Narrow limb GEMM", profiles/r05/ngemm_phase_timing_*).
The assembly is the only place where this shows without a GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
INSTANCES = {"1ELi4": 36, "2ELi2": 36, "4ELi2": 72, "8ELi1": 72}          # <KS, NT> -> matrix instructions per row tile and wave


@pytest.fixture(scope="module")
def assembly():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g.ngemm_assembly()


def kernel_body(lines, tag):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3sda23packed_gen_ngemm_kernelILi" + tag) and l.rstrip().endswith((":", ")")) or
                 l.startswith("_ZN3sda23packed_gen_ngemm_kernelILi" + tag) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops_of(body):
    """{header label: [lines of the blocks LLVM annotates as belonging to that loop]} - block labels carry `in Loop: Header=BBx_y`
    (the header itself `=>This Inner Loop Header` / `This Loop Header`), whatever order the blocks were laid out in (a rotated
    loop has its latch, with the per-tile barrier, ABOVE its header)"""
    loops, cur = {}, None
    for l in body:
        m = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)\s*(?:;\s*(.*))?$", l)
        if m:
            note = m.group(2) or ""
            if "Loop Header" in note and m.group(1):
                cur = m.group(1)
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
                cur = h.group(1) if h else None
        elif "Loop Header" in l and cur is None:
            pass
        if cur:
            loops.setdefault(cur, []).append(l)
    return loops


@pytest.mark.parametrize("tag,mfma", sorted(INSTANCES.items()))
def test_row_loop_is_free_of_scratch_accesses(assembly, tag, mfma):
    body = kernel_body(assembly, tag)
    found = False
    for header, lines in loops_of(body).items():
        code = [l.split(";")[0] for l in lines]
        if sum("v_mfma_i32_16x16x64_i8" in l for l in code) != mfma:
            continue
        found = True
        assert any("s_barrier" in l for l in code), (tag, header, "the row loop has lost its per-tile barrier")
        scratch = [l.strip() for l in code if "scratch_" in l]
        assert not scratch, (tag, "row loop touches scratch memory", scratch[:4])
        # and no vector-memory LOAD in the compute waves' loop (they never wait for vmcnt there: the loader wave streams the tiles)
        loads = [l.strip() for l in code if re.search(r"\b(global|buffer|flat)_load_(?!lds)", l)]
        assert not loads, (tag, loads[:4])
    assert found, (tag, "no loop with %d matrix instructions found" % mfma)


@pytest.mark.parametrize("ks", [1, 2, 4, 8])
def test_loader_loop_waits_with_counted_vmcnt_only(assembly, ks):
    """the loader wave's row loop (ng_loader_rows, a function of its own since round 6): tile loads straight into LDS, a counted
    s_waitcnt vmcnt, the barrier - no scratch access, no register-destination load (either would be waited for with vmcnt(0)
    and stall the tiles in flight)"""
    start = next(i for i, l in enumerate(assembly) if l.startswith("_ZN3sda14ng_loader_rowsILi%dE" % ks) and ":" in l)
    end = next(i for i in range(start, len(assembly)) if assembly[i].startswith(".Lfunc_end"))
    loops = loops_of(assembly[start:end])
    assert loops, "no loop in ng_loader_rows"
    for header, lines in loops.items():
        code = [l.split(";")[0] for l in lines]
        assert any("global_load_lds_dwordx4" in l for l in code) and any("s_barrier" in l for l in code), header
        assert not [l for l in code if "scratch_" in l or re.search(r"\b(global|buffer|flat)_load_(?!lds)", l)], header


def test_buffer_stores_are_guarded_against_the_store_hazard(assembly):
    """hipcc puts no wait state between `buffer_store_dwordx4 ..., s<N> offen` and a write to its data registers (its hazard table
    exempts MUBUF stores whose soffset is a register); on gfx950 such a store then reads the new value in 2.5 % of the cases
    (tools/microbench_store_war.hip; with soffset 0 the compiler inserts the wait state that is enough).  This cost round 5 a wrong
    4 x 4 block of shares in one launch of ten.  Since round 6 the SOURCE carries the wait state (ng_store_guard: an `s_nop 1` that
    takes the stored registers as inputs); __graft_entry__.store_hazard_findings checks both rules on the assembly, and build()
    runs the same check before it links anything."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert g.store_hazard_findings(assembly) == []
    stores = [l for l in assembly if l.strip().startswith("buffer_store_dwordx4")]
    assert stores, "the whole-tile path stores 16 bytes per lane"


def test_the_store_hazard_check_finds_what_it_is_for():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    ok = ["buffer_store_dwordx4 v[60:63], v136, s[16:19], 0 offen nt", "v_mad_i64_i32 v[0:1], s[16:17], v2, v148, 0", ";;#ASMSTART", "s_nop 1",
          "v_mov_b32 v60, v1"]
    assert g.store_hazard_findings(ok) == []
    early = ["buffer_store_dwordx4 v[60:63], v136, s[16:19], 0 offen nt", "v_mov_b32 v61, v1", "s_nop 1"]
    assert len(g.store_hazard_findings(early)) == 1
    wide = ["buffer_store_dwordx2 v[128:129], v136, s[24:27], s38 offen nt", "v_mad_i64_i32 v[128:129], s[16:17], v2, v148, 0", "s_nop 1"]
    assert len(g.store_hazard_findings(wide)) == 1
    soff = ["buffer_store_dwordx4 v[60:63], v136, s[16:19], s38 offen nt", "s_nop 1"]
    assert any("scalar offset" in f for f in g.store_hazard_findings(soff))
    unguarded = ["buffer_store_dwordx4 v[60:63], v136, s[16:19], 0 offen nt", "v_add_u32 v1, v2, v3", "s_endpgm"]
    assert len(g.store_hazard_findings(unguarded)) == 1
    assert g.store_hazard_findings(["v_add_u32 v1, v2, v3"]) != []          # no store at all: wrong file


def test_clerk_waves_load_pipeline_is_untouched_between_load_and_wait(assembly):
    """The clerk waves (ng_clerk_wave) keep TWO register sets of row loads in flight and wait for one with a counted s_waitcnt
    vmcnt(10) written by hand: the loads are inline assembly the compiler does not track.  Safe only if nothing else touches a set's
    registers between its load and the wait that names them - __graft_entry__.clerk_pipeline_findings walks the loop of every
    instance (twice: the back edge carries registers in flight); build() runs the same check."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert g.clerk_pipeline_findings(assembly) == []
    waits = [l for l in assembly if "NGCW" in l]
    assert waits and all("s_waitcnt vmcnt(10)" in l or "s_waitcnt vmcnt(0)" in l for l in waits), waits[:3]
    assert sum("s_waitcnt vmcnt(10)" in l for l in waits) >= 8                 # two per instance in the loop


def test_the_clerk_pipeline_check_finds_what_it_is_for():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    head = ["_ZN3sda23packed_gen_ngemm_kernelILi4ELi2EEEvX:", ".LBB1_1:"]
    loads = ["\tglobal_load_dwordx4 v[%d:%d], v[100:101], off nt ; NGCL" % (4 * i, 4 * i + 3) for i in range(10)]
    loads_b = ["\tglobal_load_dwordx4 v[%d:%d], v[100:101], off nt ; NGCL" % (40 + 4 * i, 43 + 4 * i) for i in range(10)]
    wait_a = "\ts_waitcnt vmcnt(10) ; NGCW " + " ".join("v[%d:%d]" % (4 * i, 4 * i + 3) for i in range(10))
    wait_b = "\ts_waitcnt vmcnt(10) ; NGCW " + " ".join("v[%d:%d]" % (40 + 4 * i, 43 + 4 * i) for i in range(10))
    tail = ["\ts_cbranch_scc1 .LBB1_1", ".Lfunc_end1:"]
    good = head + [wait_a, "\tv_add_co_u32 v110, vcc, v0, v1"] + loads + ["\ts_barrier", wait_b, "\tv_add_co_u32 v111, vcc, v40, v41"] + loads_b + ["\ts_barrier"] + tail
    assert g.clerk_pipeline_findings(good) == []
    # a copy of a register that is still in flight (set A's v[4:7] read after its load, before its wait - across the back edge)
    moved = head + [wait_a] + loads + ["\ts_barrier", wait_b, "\tv_mov_b32 v120, v5"] + loads_b + ["\ts_barrier"] + tail
    assert any("touches v5" in f for f in g.clerk_pipeline_findings(moved))
    # a use of set B right after the wait for set A
    early = head + [wait_a, "\tv_add_u32 v110, v40, v41"] + loads + ["\ts_barrier", wait_b] + loads_b + ["\ts_barrier"] + tail
    assert any("touches v40" in f for f in g.clerk_pipeline_findings(early))
    assert g.clerk_pipeline_findings(["v_add_u32 v1, v2, v3"]) != []
    assert any("missing" in f for f in g.clerk_pipeline_findings(head + ["\tv_add_u32 v1, v2, v3"] + tail))
