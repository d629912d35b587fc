# per-kernel A/B runs use the two-launch schedule (separate share-gen and clerk-sum timings)
run() { python bench.py --schedule serial --no-additional --steps 20 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-28s %.1f Gelem/s step %.2f ms gen %.2f ms comb %.2f ms verified %s path %.3f' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], k['share_gen']['avg_ms'], k['clerk_sum']['avg_ms'], d['verified_reconstruct_equals_sum'], d['path_roofline']['frac_of_hbm_peak']))" "$*"; }
run --overlap 0
run --overlap 1
run --overlap 1 --tile 1000
run --overlap 1 --tile 4000
run --overlap 1 --workload additive --steps 10
run --overlap 0 --workload additive --steps 10
