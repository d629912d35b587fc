#!/bin/bash
# the narrow limb GEMM (ngemm_kernels.hip) against the narrow transform kernel on tss's PSS_155_728_100 over tss's prime 746497,
# interleaved on one box (SDA_NO_NGEMM=1 is handed to the test-only knob by bench.py)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f launch %.3f ms verified %s kernel %s' % (d['value']/1e9, r['frac'], r.get('both_roles_launch_ms') or r.get('avg_launch_ms') or 0, d['verified_reconstruct_equals_sum'], r['kernel']))"; }
for i in 1 2; do
  echo "narrow_pss728 limb GEMM : $(fused --workload narrow_pss728 --tile 500 --participants 4000)"
  echo "narrow_pss728 transform : $(SDA_NO_NGEMM=1 fused --workload narrow_pss728 --tile 500 --participants 4000)"
done
echo "serial schedule (share generation alone, then the clerk sum):"
echo "narrow_pss728 limb GEMM : $(fused --workload narrow_pss728 --tile 500 --participants 4000 --schedule serial)"
echo "narrow_pss728 transform : $(SDA_NO_NGEMM=1 fused --workload narrow_pss728 --tile 500 --participants 4000 --schedule serial)"
