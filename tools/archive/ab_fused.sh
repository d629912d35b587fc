# interleaved A/B of the two schedules: serial (share-gen launch, then clerk-sum launch) vs fused (dual-role launch)
run() { python bench.py --steps 20 --no-cpu-baseline --no-additional "${@:2}" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %.1f Gelem/s step %.2f ms roofline %.3f (%.2f ms) ok=%s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['verified_reconstruct_equals_sum']))" "$1"; }
for i in 1 2 3; do
run "packed serial" --schedule serial
run "packed fused" --schedule fused
done
for i in 1 2; do
run "additive serial" --workload additive --steps 10 --schedule serial
run "additive fused" --workload additive --steps 10 --schedule fused
run "packed26 serial" --workload packed26 --tile 1500 --steps 10 --schedule serial
run "packed26 fused" --workload packed26 --tile 1500 --steps 10 --schedule fused
done
