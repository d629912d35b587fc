#!/bin/bash
cd "$(dirname "$0")/.."
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f launch %.3f ms verified %s' % (d['value']/1e9, r['frac'], r['both_roles_launch_ms'] or r['avg_launch_ms'], d['verified_reconstruct_equals_sum']))"; }
for i in 1 2; do
  echo "narrow_pss728 lazy   : $(fused --workload narrow_pss728 --tile 500 --participants 4000)"
  echo "narrow_pss728 reduced: $(SDA_NO_LAZY=1 fused --workload narrow_pss728 --tile 500 --participants 4000)"
done
