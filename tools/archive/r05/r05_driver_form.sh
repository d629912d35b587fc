# round 5: the driver's own command twice (compact line + details), then the same command under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_driver; mkdir -p $O
cd $R
for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/bench_details_$i.json > $O/bench_driver_form_$i.json 2> $O/bench_stderr_$i.log
  echo "run $i: rc=$? line bytes=$(wc -c < $O/bench_driver_form_$i.json)"
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --details $O/bench_details_under_rocprof.json 2>$O/rocprof.log | tail -1 > $O/bench_under_rocprof_driver_form.json
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_driver_form.csv \;
rm -rf $O/stats
head -8 $O/kernel_stats_driver_form.csv | cut -c1-200
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_driver_form_1", "bench_driver_form_2", "bench_under_rocprof_driver_form"):
    d = json.loads(open(O + "/" + f + ".json").read().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"] / 1e9, 2), round(r["frac"], 4), r["kernel"], r.get("both_roles_launch_ms"), {k: (v["frac"], v["bound"]) for k, v in d.get("additional_workloads", {}).items()})
PY
