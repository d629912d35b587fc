run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-44s %.1f Gelem/s step %.2f ms gen %.2f ms (%.2f) comb %.2f ms (%.2f) ok=%s path %.3f' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], k['share_gen']['avg_ms'], k['share_gen']['frac_of_hbm_peak'], k['clerk_sum']['avg_ms'], k['clerk_sum']['frac_of_hbm_peak'], d['verified_reconstruct_equals_sum'], d['path_roofline']['frac_of_hbm_peak']))" "$*"; }
run --workload packed --steps 20
run --workload packed_ref --steps 10
run --workload packed26 --steps 10 --tile 1500
run --workload additive --steps 10
