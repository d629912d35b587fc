#!/usr/bin/env python3
"""The measured figures of DESIGN.md / BASELINE.md / README.md, GENERATED from the records they cite - never typed by hand.

  python tools/make_design_tables.py           rewrite the blocks between `<!-- BEGIN GENERATED ... -->` / `<!-- END GENERATED -->`
  python tools/make_design_tables.py --check   exit 1 (and say where) if a block differs from what the records give

Sources: the DRIVER's latest bench record `BENCH_rNN.json` (its `parsed` line: what the judge holds), the builder's own runs of
this round's code (`BUILDER_RUNS` below: compact stdout lines kept under profiles/), `profiles/traffic.json` and
`profiles/bounds.json` (tools/make_traffic.py / make_bounds.py from the round's rocprofv3 counter passes).
tests/test_docs.py runs the check, so a figure in the three documents cannot drift from the JSON it cites."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DOCS = ["DESIGN.md", "BASELINE.md", "README.md"]
# the builder's runs of THIS round's final code (compact stdout lines of bench.py), in the order they are listed
BUILDER_RUNS = ["profiles/r06/lines/bench_line_driver_form_1.json", "profiles/r06/lines/bench_line_driver_form_2.json",
                "profiles/r06/lines/bench_line_default.json", "profiles/r06/lines/box2/bench_line_driver_form_1.json",
                "profiles/r06/lines/box2/bench_line_driver_form_2.json", "profiles/r06/lines/box2/bench_line_default.json"]
ROUND = 6

LEGS = [  # (key on the line, what it is)
    (None, "**BASELINE config 3** (headline): packed Shamir t=1 k=3 n=8, dim 1 Mi, 62-bit prime, all 100,000 participants"),
    ("packed_tss_nodes", "config 3's shape on the REFERENCE's share map (tss `share`, `packed_shamir.rs:42`)"),
    ("packed_distinct", "config 3 with every sub-tile sharing DIFFERENT participants (fill on a side stream inside the timed region; wall-clock fraction)"),
    ("additive", "BASELINE config 2: additive 3-way, 10,000 participants"),
    ("additive_chacha12", "config 2 with the CSPRNG at 12 rounds (A/B leg; the product runs ChaCha20)"),
    ("config4_full", "BASELINE config 4 at its full job size on ONE GPU: 1,000,000 participants of (k=8, t=2, n=26)"),
    ("config4_chacha12", "config 4's shape with the CSPRNG at 12 rounds (A/B leg, 50,000 participants)"),
    ("config5_full", "BASELINE config 5 at its full job size on ONE GPU: 100,000 participants at dim 16 Mi, reveal included"),
    ("narrow_ref", "tss-valid (k=3, t=4, n=8) over a 31-bit prime (the reference's own domain; narrow kernels)"),
    ("narrow26_ref", "tss-valid (k=8, t=7, n=26) over a 31-bit prime"),
    ("narrow_pss728", "tss's shipped PSS_155_728_100 over tss's prime 746497 (limb GEMM on the matrix cores + clerk waves)"),
    ("narrow_pss19682", "tss's shipped PSS_155_19682_100 over tss's prime 5038849 (limb GEMM + clerk waves)"),
    ("packed_pss728", "PSS_155_728_100 over the 62-bit prime (transform kernel, clerk sum on a side stream)"),
]


def _latest_driver_record():
    best = None
    for f in glob.glob(os.path.join(ROOT, "BENCH_r*.json")):
        m = re.search(r"BENCH_r(\d+)\.json$", f)
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        if m and isinstance(d.get("parsed"), dict) and (best is None or int(m.group(1)) > best[0]):
            # `parsed` keeps the contract's keys only; the whole stdout line (with the attached legs) is in run.stdout_tail
            line = d["parsed"]
            for cand in reversed(str((d.get("run") or {}).get("stdout_tail", "")).splitlines()):
                if cand.startswith('{"metric"'):
                    try:
                        full = json.loads(cand)
                        if abs(full["value"] - line["value"]) <= 1e-6 * line["value"]:
                            line = full
                    except ValueError:
                        pass
                    break
            d["line"] = line
            best = (int(m.group(1)), os.path.basename(f), d)
    return best


def _leg(line, key):
    if key is None:
        return {"value": line["value"], "frac": line["roofline"]["frac"], "bound": line["roofline"].get("bound"),
                "verified": line.get("verified_reconstruct_equals_sum")}
    return (line.get("additional_workloads") or {}).get(key)


def _fmt_leg(leg):
    if not leg:
        return "- (leg not on that line)"
    return "%.1f = %.3f" % (leg["value"] / 1e9, leg["frac"]) if leg["value"] >= 1e10 else "%.2f = %.3f" % (leg["value"] / 1e9, leg["frac"])


def bench_block():
    drv = _latest_driver_record()
    runs = [(p, json.load(open(os.path.join(ROOT, p)))) for p in BUILDER_RUNS if os.path.exists(os.path.join(ROOT, p))]
    out = []
    if drv:
        n, name, d = drv
        stale = n < ROUND
        out.append(f"Driver record: `{name}` (`{d.get('cmd')}`, head `{str(d.get('head'))[:7]}`, `driver_run_s` {d.get('driver_run_s')})"
                   + (f" - **round-{n} code**: the driver has not yet run round {ROUND}'s; legs that round {ROUND} changed "
                      f"(`narrow_pss728`, `narrow_pss19682`) and legs it added are marked" if stale else "") + ".")
    else:
        out.append("Driver record: none with a parsed line.")
    out.append(f"Builder runs of round {ROUND}'s code (each on a fresh gpurun box; the same command gives figures 2 - 4 % apart from run to run "
               f"and box to box - `profiles/r06/headline_warmup.txt`, `headline_launch_pattern.txt`, `headline_arena.txt`): "
               + ", ".join(f"`{p}` (`--steps {l['steps']} --warmup {l['warmup']}`, build `{l.get('build_id')}`)" for p, l in runs) + ".")
    out.append("")
    out.append("| leg of the bench line | driver: G elements/s = fraction of 8 TB/s | builder boxes, round %d code: G elements/s = fraction (lo - hi) | bound | verified |" % ROUND)
    out.append("|---|---|---|---|---|")
    for key, what in LEGS:
        dleg = _leg(drv[2]["line"], key) if drv else None
        bl = [x for x in (_leg(l, key) for _, l in runs) if x]
        if bl:
            vals, fracs = [x["value"] / 1e9 for x in bl], [x["frac"] for x in bl]
            fv = "%.1f" if max(vals) >= 10 else "%.2f"
            b = (fv + " - " + fv + " = %.3f - %.3f") % (min(vals), max(vals), min(fracs), max(fracs)) if len(bl) > 1 else _fmt_leg(bl[0])
            extra = ""
            if key == "packed_distinct" and bl[0].get("frac_with_fill"):
                extra = "; with the fill's 8 B/element counted: %.3f - %.3f" % (min(x["frac_with_fill"] for x in bl), max(x["frac_with_fill"] for x in bl))
            bound = bl[-1].get("bound")
            ver = all(x.get("verified") for x in bl)
        else:
            b, extra, bound, ver = "-", "", None, None
        out.append("| %s%s | %s | %s%s | %s | %s |" % (what, f" (`{key}`)" if key else "", _fmt_leg(dleg), b, extra,
                                                     bound if bound else "not claimed", "yes" if ver else "-" if ver is None else "**NO**"))
    if runs:
        c = runs[0][1].get("cpu_baseline") or {}
        if c:
            ac = c.get("all_cores") or {}
            out.append("| CPU port (`oracle/sda_oracle.c`) on the same box: %s | %s | %.4f on one core%s | - | - |" % (
                c.get("cpu_model"), "%.4f on one core" % (drv[2]["line"]["cpu_baseline"]["value"] / 1e9) if drv and drv[2]["line"].get("cpu_baseline") else "-",
                c["value"] / 1e9, "; %.3f best of the thread sweep (%s threads%s)" % (ac["value"] / 1e9, ac.get("cores"), ", cached sweep" if ac.get("cached") else "") if ac else ""))
    return "\n".join(out)


def counters_block():
    import bench
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    bounds = json.load(open(os.path.join(ROOT, "profiles", "bounds.json")))
    import __graft_entry__ as ge
    now = ge.kernel_digest()
    stamps = {traffic.get("_kernel_id"), bounds.get("_kernel_id")}
    state = (f"measured on the kernels of this tree (`sda_kernel_id()` = `{now}`)" if stamps == {now} else
             f"**STALE: measured on kernel id {sorted(str(x) for x in stamps)}, the tree's device code is `{now}` - rerun `tools/profile_r06.sh`, "
             f"`make_traffic.py`, `make_bounds.py`**")
    out = [f"Tables {state}; `bench.py` prints the same comparison as `roofline.tables_current`.", "",
           "Counters (`profiles/traffic.json`, `profiles/bounds.json`; written by `tools/make_traffic.py` / `tools/make_bounds.py` from the "
           "rocprofv3 passes of `tools/profile_r06.sh`: kernel stats, FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes, SQ counters; "
           "FETCH_SIZE doubled as the microarchitecture guide prescribes).  One launch = one tile of the size shown; \"algorithmic\" = "
           "tile x dim x (8 + 16 n / k) B, for the limb GEMM plus the clerk sum's 128-bit running sums (n x B x 32 B read and written per launch, "
           "the price of a 40- or 500-participant tile).", "",
           "| workload (tile) | launch | kernel | bound | PMC GB / launch | algorithmic GB | ratio | PMC GB/s (of the no-arithmetic floor) | VALU wave-instr / element | VALU busy |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(k for k in traffic if not k.startswith("_")):
        name, tile, dim = key.split(":")
        P, dim = int(tile[4:]), int(dim[3:])
        w = bench.WORKLOADS[name]
        n, k = w["n"], w["k"]
        gen_b, comb_b = 8 + 8 * n / k, 8 * n / k
        B = -(-dim // k)
        state = 32.0 * n * B if name.startswith("narrow_pss") else 0.0
        e, bd = traffic[key], bounds.get(name, {})
        for role, tkey, alg in (("fused", "fused_bytes_per_launch", P * dim * (gen_b + comb_b) + state),
                                ("serial_gen", "gen_bytes_per_launch", P * dim * gen_b),
                                ("serial_comb", "comb_bytes_per_launch", P * dim * comb_b + state)):
            if tkey not in e or role not in bd:
                continue
            if role == "serial_comb" and "fused" in bd and "fused_bytes_per_launch" in e:
                continue                                             # the last launch's plain clerk sum beside a dual-role kernel: not a row
            b = bd[role]
            h, v = b.get("hbm") or {}, b.get("valu") or {}
            kern = (e.get(tkey.replace("_bytes_per_launch", "_kernel")) or v.get("kernel", "")).replace("void ", "").replace("sda::", "")
            out.append("| %s (%d) | %s | `%s` | **%s** | %.1f | %.1f | %.3f | %s | %.2f | %.2f |" % (
                name, P, {"fused": "dual-role", "serial_gen": "share-gen", "serial_comb": "clerk-sum"}[role], kern, b.get("bound"),
                e[tkey] / 1e9, alg / 1e9, e[tkey] / alg, "%.0f (%.2f)" % (h["pmc_GBps"], h["frac_of_floor"]) if h else "-",
                v.get("valu_wave_instr_per_element", float("nan")), v.get("valu_busy", float("nan"))))
    return "\n".join(out)


BLOCKS = {"bench": bench_block, "counters": counters_block}


def render(text):
    def sub(m):
        return f"<!-- BEGIN GENERATED {m.group(1)} (tools/make_design_tables.py) -->\n{BLOCKS[m.group(1)]()}\n<!-- END GENERATED -->"
    return re.sub(r"<!-- BEGIN GENERATED (\w+) \(tools/make_design_tables\.py\) -->\n(?:(?!<!-- BEGIN GENERATED).)*?<!-- END GENERATED -->", sub, text, flags=re.S)


def main():
    check = "--check" in sys.argv
    bad = []
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        old = open(path).read()
        new = render(old)
        if "BEGIN GENERATED" not in old:
            bad.append(f"{doc}: no generated block")
        elif new != old:
            if check:
                bad.append(f"{doc}: a generated block differs from the records (run python tools/make_design_tables.py)")
            else:
                open(path, "w").write(new)
                print("updated", doc)
    if bad:
        print("\n".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
