# A/B of the limb-GEMM share-gen kernel (SDA_FORCE_MFMA=1) against the limb-31 kernel: default dual-role schedule and share-gen alone
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f both-roles launch %.3f ms' % (d['value']/1e9, r['frac'], r['both_roles_launch_ms']))"; }
serial() { python bench.py --schedule serial --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%.1f Gelem/s gen %.3f ms comb %.3f ms' % (d['value']/1e9, k['share_gen']['avg_ms'], k['clerk_sum']['avg_ms']))"; }
for w in ${WL:-packed26_ref packed26}; do
  for i in 1 2; do
    echo "$w dual-role l31 : $(fused --workload $w)"
    echo "$w dual-role mfma: $(SDA_FORCE_MFMA=1 fused --workload $w)"
  done
  echo "$w serial l31 : $(serial --workload $w)"
  echo "$w serial mfma: $(SDA_FORCE_MFMA=1 serial --workload $w)"
done
