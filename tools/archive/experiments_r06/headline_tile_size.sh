cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('tile %d x %d sub-tiles, %d steps: %.1f Gelem/s frac %.4f both-roles %.3f ms = %.3f us/participant' % (c['tile_participants'], c['sub_tiles_per_step'], d['steps'], d['value']/1e9, r['frac'], r['both_roles_launch_ms'], 1e3*r['both_roles_launch_ms']/c['tile_participants']))"; }
for rep in 1 2 3; do
  run --steps 20 --warmup 5
  run --steps 20 --warmup 5 --tile 2000
  run --steps 25 --warmup 5 --tile 2000
  run --steps 20 --warmup 5 --tile 1000
  run --steps 50 --warmup 3
done
rocm-smi --showmeminfo vram 2>/dev/null | head -8
