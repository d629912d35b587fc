cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s mean %.3f:' % (d['value']/1e9, r['both_roles_launch_ms']), ' '.join('%.1f' % x for x in r['launch_ms_each']))"
done
