# round-2 first GPU pass: suite, default bench line, evidence for the reference-valid shapes (kernel stats + HBM PMC)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_first; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_err.txt | tail -1 > $O/bench_driver_form.json; head -c 2500 $O/bench_driver_form.json; echo
for wl in packed_ref packed26_ref; do
  timeout 900 bash tools/profile_workload.sh $wl --workload $wl --participants 15000 --tile 1500 2>&1 | tail -25
  mkdir -p $O/$wl; cp -r gpurun_out/prof_$wl/*.json $O/$wl/ 2>/dev/null; find gpurun_out/prof_$wl/stats -name '*kernel_stats.csv' -exec cp {} $O/$wl/kernel_stats.csv \;
done
