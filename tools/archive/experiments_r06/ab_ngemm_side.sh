#!/bin/bash
# limb GEMM + clerk sum: the dual-role grid against the two launches side by side with single-wave clerk workgroups on the side
# stream (knob SDA_NGEMM_SIDE_WAVES = such workgroups per CU; SDA_SIDE_STREAM_PRIORITY=1: the side stream at HIGH priority)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f Gelem/s frac %.3f launch %.3f ms verified %s kernel %s' % (d['value']/1e9, r['frac'], r.get('both_roles_launch_ms') or r.get('avg_launch_ms') or 0, d['verified_reconstruct_equals_sum'], r['kernel']))"; }
for wl in "narrow_pss728 --tile 500 --participants 3000" "narrow_pss19682 --tile 40 --participants 240"; do
  echo "== $wl"
  echo "dual-role grid        : $(fused --workload $wl)"
  for w in 3 6 12; do
    echo "side waves $w (low)    : $(SDA_NGEMM_SIDE_WAVES=$w fused --workload $wl)"
  done
  echo "side waves 6 (high)   : $(SDA_NGEMM_SIDE_WAVES=6 SDA_SIDE_STREAM_PRIORITY=1 fused --workload $wl)"
done
