# the driver's own command under rocprofv3 --kernel-trace --stats (kernel durations to hold against the line's HIP-event figures)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03_driver; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --full-line --steps 20 --warmup 5 --no-cpu-baseline 2>$O/rocprof.log | tail -1 > $O/bench_under_rocprof_driver_form.json
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_driver_form.csv \;
rm -rf $O/stats
head -6 $O/kernel_stats_driver_form.csv | cut -c1-220
cd $R && for i in 1 2; do python bench.py --full-line --steps 20 --warmup 5 --no-cpu-baseline --no-additional 2>/dev/null | tail -1 > $O/bench_driver_form_repeat_$i.json; done
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_under_rocprof_driver_form", "bench_driver_form_repeat_1", "bench_driver_form_repeat_2"):
    d = json.loads(open(O + "/" + f + ".json").read().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"] / 1e9, 2), round(r["frac"], 4), r["avg_launch_ms"], r["both_roles_launch_ms"], r["launches"])
PY
