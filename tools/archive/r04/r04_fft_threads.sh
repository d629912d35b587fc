#!/bin/bash
cd "$(dirname "$0")/.."
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f launch %.3f ms verified %s' % (d['value']/1e9, r['frac'], r['both_roles_launch_ms'] or r['avg_launch_ms'], d['verified_reconstruct_equals_sum']))"; }
for th in 512 384 448 576 640 704 768; do
  echo "narrow_pss728 threads=$th: $(SDA_FFT_THREADS=$th fused --workload narrow_pss728 --tile 500 --participants 4000)"
done
for th in 512 448 384 576; do
  echo "packed_pss728 (62-bit) threads=$th: $(SDA_FFT_THREADS=$th fused --workload packed_pss728 --tile 500 --participants 4000)"
done
