cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('%-14s tile %4d: %.1f Gelem/s frac %.4f both-roles %.3f ms' % (c['name'], c['tile_participants'], d['value']/1e9, r['frac'], r['both_roles_launch_ms']))"; }
for rep in 1 2; do
for w in narrow26_ref narrow_ref packed26 additive; do
for t in 1000 1250 1500 2000 2500; do
  run --workload $w --steps 8 --warmup 2 --tile $t --participants $((8*t))
done; done; done
