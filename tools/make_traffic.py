#!/usr/bin/env python3
"""profiles/rNN/final/<workload>/{pmc_hbm.json, bench_under_rocprof.json} -> profiles/traffic.json (HBM bytes per launch of the
dominant kernel, FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on
gfx950, WRITE_SIZE as reported) and a summary table on stdout.   python tools/make_traffic.py profiles/r04/final"""
import json, os, sys

root = sys.argv[1]
traffic = {"_note": "HBM bytes per launch from rocprofv3 PMC (separate --pmc passes, kernel-trace only; tools/profile_r04.sh). "
                    "FETCH_SIZE (KiB) is doubled as the microarchitecture guide prescribes for wide coalesced 16 B/lane reads on "
                    "gfx950; WRITE_SIZE (KiB) is used as reported (calibrated in round 1 on fill_synthetic_kernel: 16.777 GB written, "
                    "16.777 GB reported). Dual-role kernels: the both-roles launches (the largest grid)."}
rows = []
kernel_ids = set()
for w in sorted(os.listdir(root)):
    d = os.path.join(root, w)
    if not os.path.isfile(os.path.join(d, "pmc_hbm.json")):
        continue
    pmc = json.load(open(os.path.join(d, "pmc_hbm.json")))
    line = json.loads(open(os.path.join(d, "bench_under_rocprof.json")).read().splitlines()[-1])
    kernel_ids.add(line.get("kernel_id"))
    cfg = line["config"]
    P, dim = cfg["tile_participants"], cfg["dim"]
    per = {}
    for key, v in pmc.items():
        kern, counter, grid = [x.strip() for x in key.split("::", 1)[0:1]] + [x.strip() for x in key.rsplit("::", 2)[1:]]
        kern = key.rsplit("::", 2)[0].strip()
        per.setdefault(kern, {}).setdefault(grid, {})[counter] = v
    entry = {}
    for kern, grids in per.items():
        # the steady-state grid: the largest one (both roles for the dual-role kernels: n_gen + n_comb workgroups)
        grid, c = max(grids.items(), key=lambda kv: int(kv[0].split()[-1]))
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fetch = c["FETCH_SIZE"]["mean_KiB"] * 1024 * 2
        write = c["WRITE_SIZE"]["mean_KiB"] * 1024
        short = kern.replace("void ", "").split("<")[0].replace("sda::", "")
        dual_ngemm = short == "packed_gen_ngemm_kernel" and str(cfg.get("schedule", "")).startswith("dual-role")
        tag = ("fused" if "fused" in short or dual_ngemm else "rest" if short == "ngemm_clerk_rest_kernel" else
               "gen" if "gen" in short else "comb" if "combine_update" in short else None)
        if tag:
            entry[tag + "_kernel"] = kern
            entry[tag + "_fetch_corrected_bytes"] = fetch
            entry[tag + "_write_bytes"] = write
            entry[tag + "_bytes_per_launch"] = fetch + write
    if "rest_bytes_per_launch" in entry and "fused_bytes_per_launch" in entry:       # one call = the dual-role kernel + its follow-up kernel
        entry["fused_bytes_per_launch"] += entry["rest_bytes_per_launch"]
    traffic[f"{cfg['name']}:tile{P}:dim{dim}"] = entry
    n, k = cfg["share_count"], cfg["secret_count"]
    alg = P * dim * (8 + 16 * n / k)
    r = line["roofline"]
    meas = entry.get("fused_bytes_per_launch") or (entry.get("gen_bytes_per_launch", 0) + entry.get("comb_bytes_per_launch", 0))
    rows.append((cfg["name"], P, line["value"] / 1e9, line["path_roofline"]["frac_of_hbm_peak"], r["kernel"], r["avg_launch_ms"], alg / 1e9, meas / 1e9))
# the kernels the passes ran on (sda_kernel_id() of the profiled build, from the records): ONE id or the table says "mixed"
traffic["_kernel_id"] = kernel_ids.pop() if len(kernel_ids) == 1 else "mixed: " + ", ".join(sorted(str(k) for k in kernel_ids))
json.dump(traffic, open(os.path.join(os.path.dirname(root.rstrip("/")), "traffic.json"), "w"), indent=1)
# ... and the copy bench.py reads for `roofline.traffic` (always the current round's)
json.dump(traffic, open(os.path.join(os.path.dirname(os.path.dirname(root.rstrip("/"))), "traffic.json"), "w"), indent=1)
print("| workload | tile | Gelem/s | path frac of 8 TB/s | dominant kernel | avg launch ms | algorithmic GB/launch | PMC GB/launch |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %d | %.1f | %.3f | %s | %.2f | %.1f | %.1f |" % r)
