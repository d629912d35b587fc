run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-52s %.1f Gelem/s step %.2f ms gen %.2f ms (%.2f) comb %.2f ms (%.2f) ok=%s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], k['share_gen']['avg_ms'], k['share_gen']['frac_of_hbm_peak'], k['clerk_sum']['avg_ms'], k['clerk_sum']['frac_of_hbm_peak'], d['verified_reconstruct_equals_sum']))" "$*"; }
run --workload packed --dim 16777216 --tile 128 --steps 10
run --workload packed --dim 16777216 --tile 256 --steps 5
run --workload packed --dim 1048576 --tile 4000 --steps 5
run --workload packed --dim 1000003 --tile 777 --steps 5
run --workload additive --dim 1000003 --tile 777 --steps 5
run --workload packed26 --dim 16777216 --tile 96 --steps 3
