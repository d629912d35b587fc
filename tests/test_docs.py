"""The documents quote what the records hold: every measured figure of DESIGN.md, BASELINE.md and README.md
sits in a block that tools/make_design_tables.py generates from BENCH_rNN.json (the driver's record), the builder's kept bench
lines and profiles/{traffic,bounds}.json.  A block that differs from what those files give fails here."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_blocks_are_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_design_tables.py"), "--check"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr


def test_the_blocks_quote_the_driver_record_and_the_counter_files():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_design_tables as m
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    blocks = dict(re.findall(r"<!-- BEGIN GENERATED (\w+) \(tools/make_design_tables\.py\) -->\n(.*?)<!-- END GENERATED -->", design, flags=re.S))
    assert set(blocks) == {"bench", "counters"}
    n, name, d = m._latest_driver_record()
    line = d["line"]
    # the headline cell of the driver column is the driver's number, to the digit
    assert "| %.1f = %.3f |" % (line["value"] / 1e9, line["roofline"]["frac"]) in blocks["bench"] and name in blocks["bench"]
    for key, leg in (line.get("additional_workloads") or {}).items():
        cell = ("%.1f = %.3f" if leg["value"] >= 1e10 else "%.2f = %.3f") % (leg["value"] / 1e9, leg["frac"])
        assert cell in blocks["bench"], (key, cell)
    # every counter row is the JSON's figure
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, e in traffic.items():
        if key.startswith("_"):
            continue
        got = e.get("fused_bytes_per_launch") or e.get("gen_bytes_per_launch")
        assert "| %.1f |" % (got / 1e9) in blocks["counters"], (key, got)
    for doc in ("BASELINE.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert blocks["bench"] in text, doc + " carries another bench table than DESIGN.md"


def test_prose_carries_no_throughput_figures_of_its_own():
    """outside the generated blocks the three documents state no 'Gelem/s = fraction' pairs (the round-5 review found ranges that
    excluded the driver's record); historical figures live in CHANGELOG.md"""
    for doc in ("DESIGN.md", "BASELINE.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        text = re.sub(r"<!-- BEGIN GENERATED.*?<!-- END GENERATED -->", "", text, flags=re.S)
        hits = re.findall(r"\d{2,3}(?:\.\d)?\s*(?:-|–)?\s*(?:\d{2,3}(?:\.\d)?)?\s*G(?:elem| elements)/s\s*=\s*0\.\d+", text)
        assert not hits, (doc, hits[:5])
