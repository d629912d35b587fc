#!/bin/bash
# the clerk sum beside the narrow limb GEMM: walk workgroups and priority of the side stream (knobs SDA_SIDE_STREAM_WGS, SDA_SIDE_STREAM_PRIORITY)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f launch %.3f ms verified %s' % (d['value']/1e9, r['frac'], r.get('both_roles_launch_ms') or r.get('avg_launch_ms') or 0, d['verified_reconstruct_equals_sum']))"; }
for pr in l h; do for w in 256 512 1024 2048 4096; do
  echo "wgs $w priority $pr: $(SDA_SIDE_STREAM_WGS=$w SDA_SIDE_STREAM_PRIORITY=$pr fused --workload narrow_pss728 --tile 500 --participants 4000)"
done; done
echo "no side stream: $(SDA_NO_SIDE_STREAM=1 fused --workload narrow_pss728 --tile 500 --participants 4000)"
