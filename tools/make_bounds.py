#!/usr/bin/env python3
"""profiles/rNN/final/<workload>/{pmc_hbm.json, pmc_sq.json, bench_under_rocprof.json}  ->  profiles/bounds.json:
which ceiling is active for each bench workload's dominant kernel (SURVEY.md 8d "report which bound is active"), read
by bench.py for `roofline.bound`.     python tools/make_bounds.py profiles/r04/final

Rule.  "hbm": the HBM bytes the PMC counters saw per launch / the launch's duration (HIP events of the same command) is
within 10 % of what tools/microbench_hbm reaches with the SAME access pattern and no arithmetic at all (the floor).
Otherwise "valu" when rocprof's VALUBusy is at least 0.75 - the launch is limited by vector-ALU instruction issue - and
"neither" below that (round 5: e.g. the narrow limb GEMM, whose vector and matrix instructions take turns on
the SIMD); the record carries the numbers to recompute either:
  valu_wave_instr_per_element  = SQ_INSTS_VALU / (participants x dim of one launch)
  simd_cycles_per_valu_instr   = (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs / SQ_INSTS_VALU   (SIMD cycles available per issued
                                 VALU wave-instruction; tools/microbench_valu: 2.4-2.8 for add/sub/xor/and/mov, 4.1-5.3
                                 for every other integer instruction, so ~3.7-4 for these mixes means saturated ALUs)
  valu_busy                    = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)   (rocprof's VALUBusy)
"""
import json, os, sys

root = sys.argv[1]
# GB/s tools/microbench_hbm moves with each access pattern and NO arithmetic (DESIGN.md 5 "Schedules", profiles/r01)
FLOOR = {"fused": 5940.0, "serial_gen": 6140.0, "serial_comb": 6510.0}
XCDS, SIMDS = 8, 1024
bounds = {"_note": __doc__.split("Rule.")[1].strip(), "_floors_GBps": FLOOR}
rows = []
kernel_ids = set()
for w in sorted(os.listdir(root)):
    d = os.path.join(root, w)
    if not os.path.isfile(os.path.join(d, "pmc_sq.json")):
        continue
    line = json.loads(open(os.path.join(d, "bench_under_rocprof.json")).read().splitlines()[-1])
    kernel_ids.add(line.get("kernel_id"))
    cfg, roof = line["config"], line["roofline"]
    P, dim = cfg["tile_participants"], cfg["dim"]
    hbm = json.load(open(os.path.join(d, "pmc_hbm.json")))
    sq = json.load(open(os.path.join(d, "pmc_sq.json")))
    per = {}
    for key, v in hbm.items():
        kern = key.rsplit("::", 2)[0].strip()
        counter, grid = [x.strip() for x in key.rsplit("::", 2)[1:]]
        per.setdefault(kern, {}).setdefault(grid, {})[counter] = v["mean_KiB"] * 1024
    entry = {}
    for key, c in sq.items():
        kern, grid = [x.strip() for x in key.rsplit("::", 1)]
        short = kern.replace("void ", "").split("<")[0].replace("sda::", "")
        dual_ngemm = short == "packed_gen_ngemm_kernel" and str(cfg.get("schedule", "")).startswith("dual-role")
        role = ("fused" if short.startswith("fused") or dual_ngemm else "serial_gen" if "_gen_" in short or short.endswith("gen_kernel") else
                "serial_comb" if short == "combine_update_kernel" else None)
        if role is None or "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        # the steady-state launch: the largest grid of that kernel (both roles for the dual-role kernels)
        if role in entry and int(grid.split()[-1]) <= entry[role]["_grid"]:
            continue
        if role == "fused":
            ms = roof.get("both_roles_launch_ms")
        elif "kernels" in line:
            ms = line["kernels"]["share_gen" if role == "serial_gen" else "clerk_sum"]["avg_ms"]
        else:
            ms = None
        traffic = per.get(kern, {}).get(grid, {})
        pmc_bytes = 2 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"] if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic else None
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        valu = {"kernel": kern, "valu_wave_instr_per_launch": c["SQ_INSTS_VALU"],
                "valu_wave_instr_per_element": c["SQ_INSTS_VALU"] / (P * dim),
                "simd_cycles_per_valu_instr": cyc * SIMDS / c["SQ_INSTS_VALU"],
                "valu_busy": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / SIMDS / cyc,
                "gpu_cycles_per_launch": cyc, "elements_per_launch": P * dim}
        e = {"_grid": int(grid.split()[-1]), "valu": valu}
        if pmc_bytes and ms:
            gbps = pmc_bytes / (ms * 1e-3) / 1e9
            e["hbm"] = {"pmc_bytes_per_launch": pmc_bytes, "launch_ms": ms, "pmc_GBps": gbps, "floor_GBps": FLOOR[role],
                        "frac_of_floor": gbps / FLOOR[role]}
            # round 5: a third verdict - neither ceiling is reached (HBM below 0.9 of its floor AND the vector ALUs busy less than
            # 0.75 of the time): the launch is limited by something else (vector and matrix instructions taking turns on the SIMD in
            # the narrow limb GEMM; the store stream of its 19682-clerk shape) and saying "valu" would be a claim without evidence
            e["bound"] = "hbm" if gbps >= 0.9 * FLOOR[role] else "valu" if valu["valu_busy"] >= 0.75 else "neither"
        else:
            e["bound"] = "valu" if valu["valu_busy"] > 0.6 else "hbm"
        e["evidence"] = f"{root.rstrip('/')}/{w}/{{pmc_sq.json,pmc_hbm.json,bench_under_rocprof.json}} (tile {P}, dim {dim}); rule in profiles/bounds.json _note"
        entry[role] = e
    for e in entry.values():
        e.pop("_grid", None)
    bounds[cfg["name"]] = entry
    for role, e in entry.items():
        h = e.get("hbm", {})
        rows.append((cfg["name"], role, e["bound"], h.get("pmc_GBps", float("nan")), h.get("frac_of_floor", float("nan")),
                     e["valu"]["valu_wave_instr_per_element"], e["valu"]["simd_cycles_per_valu_instr"], e["valu"]["valu_busy"]))
bounds["_kernel_id"] = kernel_ids.pop() if len(kernel_ids) == 1 else "mixed: " + ", ".join(sorted(str(k) for k in kernel_ids))
out = os.path.join(os.path.dirname(os.path.dirname(root.rstrip("/"))), "bounds.json")
json.dump(bounds, open(out, "w"), indent=1)
print("| workload | launch | bound | PMC GB/s | of the no-arithmetic floor | VALU wave-instr / element | SIMD cycles / VALU instr | VALU busy |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %s | **%s** | %.0f | %.2f | %.2f | %.2f | %.2f |" % r)
