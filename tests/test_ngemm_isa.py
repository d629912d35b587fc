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


@pytest.mark.parametrize("tag,mfma", sorted(INSTANCES.items()))
def test_row_loop_is_free_of_scratch_accesses(assembly, tag, mfma):
    body = kernel_body(assembly, tag)
    # the row loop: from the header of an innermost loop to the per-tile s_barrier that follows the row tile's matrix instructions
    headers = [i for i, l in enumerate(body) if "Loop Header" in l] + [len(body)]
    found = False
    for h, nxt in zip(headers, headers[1:]):
        seg = [l.split(";")[0] for l in body[h:nxt]]
        count, stop = 0, None
        for i, l in enumerate(seg):
            count += "v_mfma_i32_16x16x64_i8" in l
            if count == mfma and "s_barrier" in l:
                stop = i
                break
        if stop is None or count != mfma:
            continue
        found = True
        loop = seg[:stop + 1]
        scratch = [l.strip() for l in loop if "scratch_" in l]
        assert not scratch, (tag, "row loop touches scratch memory", scratch[:4])
        # and no vector-memory LOAD in the compute waves' loop (the global_load_lds prefetches have no destination register and are
        # never waited for)
        loads = [l.strip() for l in loop if re.search(r"\b(global|buffer|flat)_load_(?!lds)", l)]
        assert not loads, (tag, loads[:4])
    assert found, (tag, "no loop with %d matrix instructions found" % mfma)



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
