# DRBG round count (SDA_DRBG_ROUNDS = 20 default / 12 / 8) on the default dual-role schedule
run() { SDA_DRBG_ROUNDS=$1 python bench.py --steps 10 --no-cpu-baseline --no-additional "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.1f Gelem/s step %.2f ms ok=%s %s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], d['verified_reconstruct_equals_sum'], d['config']['randomness']))" "rounds=$1 ${*:2}"; }
for w in packed additive packed_ref; do for r in 20 12 8; do run $r --workload $w; done; done
