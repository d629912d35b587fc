#!/bin/bash
# interleaved A/B on one box: the systematic CSPRNG share map (default) against tss's nodes (SDA_BENCH_SHARE_MAP=tss), dual-role launch
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f both-roles launch %.3f ms verified %s' % (d['value']/1e9, r['frac'], r['both_roles_launch_ms'], d['verified_reconstruct_equals_sum']))"; }
for w in ${WL:-packed26 packed_ref packed26_ref packed additive}; do
  T=2500; [ $w = packed26 ] && T=1250; [ $w = packed26_ref ] && T=1500; [ $w = packed_ref ] && T=1500; [ $w = additive ] && T=2000
  for i in 1 2; do
    echo "$w systematic: $(fused --workload $w --tile $T --participants $((8*T)))"
    echo "$w tss-nodes : $(SDA_BENCH_SHARE_MAP=tss fused --workload $w --tile $T --participants $((8*T)))"
  done
done
