run() { python bench.py --steps 20 --no-cpu-baseline --no-verify "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-34s %.1f Gelem/s step %.2f gen %.2f comb %.2f' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], k['share_gen']['avg_ms'], k['clerk_sum']['avg_ms']))" "$1"; }
run "serial" --overlap 0
for pad in 160000 80000 53000 40000; do
SDA_COMB_LDS_PAD=$pad run "serial comb_pad$pad" --overlap 0
SDA_COMB_LDS_PAD=$pad run "overlap comb_pad$pad" --overlap 1
done
