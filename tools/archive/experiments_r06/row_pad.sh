cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('%-13s pad %5s stride %7d: %.1f Gelem/s frac %.4f both-roles %.3f ms' % (c['name'], os.environ.get('SDA_BENCH_ROW_PAD','0'), c['row_stride_elements'], d['value']/1e9, r['frac'], r['both_roles_launch_ms']))"; }
for rep in 1 2; do
for w in "packed26 --tile 1250 --participants 25000 --steps 20" "narrow26_ref --tile 2000 --participants 24000 --steps 12"; do
for pad in 0 16 48 112 240 1040 4112; do
  SDA_BENCH_ROW_PAD=$pad run --workload $w --warmup 2
done; done; done
