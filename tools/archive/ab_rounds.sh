# CSPRNG round count (--drbg-rounds 20 default / 12 / 8; A/B only - the product runs ChaCha20) on the default dual-role schedule
run() { python bench.py --steps 10 --no-cpu-baseline --no-additional --drbg-rounds $1 "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.1f Gelem/s step %.2f ms ok=%s %s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], d['verified_reconstruct_equals_sum'], d['config']['randomness']))" "rounds=$1 ${*:2}"; }
for r in 20 12 8; do run $r "$@"; done
