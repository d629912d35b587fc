# per-kernel A/B runs use the two-launch schedule (separate share-gen and clerk-sum timings)
run() { python bench.py --schedule serial --steps 20 --no-cpu-baseline --no-verify --no-additional "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-28s %.1f Gelem/s step %.2f gen %.2f comb %.2f' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], k['share_gen']['avg_ms'], k['clerk_sum']['avg_ms']))" "$1"; }
for r in 20 12 8; do SDA_DRBG_ROUNDS=$r run "packed rounds=$r"; done
for r in 20 12 8; do SDA_DRBG_ROUNDS=$r run "additive rounds=$r" --workload additive --steps 10; done
