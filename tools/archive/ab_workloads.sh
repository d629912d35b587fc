# every BASELINE shape and its tss-valid neighbour, both schedules
run() { python bench.py --no-cpu-baseline --no-additional "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-58s %.1f Gelem/s step %.2f ms path %.3f of peak ok=%s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], d['path_roofline']['frac_of_hbm_peak'], d['verified_reconstruct_equals_sum']))" "$*"; }
for s in fused serial; do
run --workload packed --steps 20 --schedule $s
run --workload packed_ref --steps 10 --schedule $s
run --workload packed26 --steps 10 --tile 1500 --schedule $s
run --workload packed26_ref --steps 10 --tile 1500 --schedule $s
run --workload additive --steps 10 --schedule $s
done
