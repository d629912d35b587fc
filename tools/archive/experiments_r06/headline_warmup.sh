cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('warmup %3d steps %d tile %d: %.1f Gelem/s frac %.4f both-roles mean %.3f median %.3f min/max %s' % (d['warmup'], d['steps'], c['tile_participants'], d['value']/1e9, r['frac'], r['both_roles_launch_ms'], r['both_roles_launch_ms_median'], [round(x,2) for x in r['both_roles_launch_ms_min_max']]))"; }
for rep in 1 2 3; do
  run --steps 20 --warmup 5
  run --steps 20 --warmup 40
  run --steps 20 --warmup 100
done
