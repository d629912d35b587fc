cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('tile %d: %.1f Gelem/s frac %.4f both-roles %.3f ms = %.3f us/participant' % (c['tile_participants'], d['value']/1e9, r['frac'], r['both_roles_launch_ms'], 1e3*r['both_roles_launch_ms']/c['tile_participants']))"; }
for rep in 1 2; do
for t in 1000 1250 1500 1750 2000 2048 2250 2500 3000; do
  run --steps 12 --warmup 3 --tile $t --participants $((12*t))
done; done
