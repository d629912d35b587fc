# config 4's wide group: Karatsuba form (default) against the plain form (knob SDA_NO_KARATSUBA=1), interleaved
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line --workload packed26 --tile 1250 --participants 50000 --steps 20 --warmup 2 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s %.1f Gelem/s frac %.4f both-roles %.3f ms verified %s' % ('plain' if os.environ.get('SDA_NO_KARATSUBA') else 'karatsuba', d['value']/1e9, r['frac'], r['both_roles_launch_ms'], d['verified_reconstruct_equals_sum']))"; }
for rep in 1 2 3; do
  run
  SDA_NO_KARATSUBA=1 run
done
