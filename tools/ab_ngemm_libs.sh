#!/bin/bash
# interleaved A/B of limb GEMM variants (bash tools/build_kernel_variant.sh ngemm_kernels.hip NAME [-D...]): usage  bash tools/ab_ngemm_libs.sh NAME [NAME ...]
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f Gelem/s frac %.3f launch %.3f ms first(gen only) %.3f last(clerk only) %.3f verified %s' % (d['value']/1e9, r['frac'], r.get('both_roles_launch_ms') or 0, r.get('first_launch_ms_share_gen_only') or 0, r.get('last_launch_ms_clerk_sum_only') or 0, d['verified_reconstruct_equals_sum']))"; }
for rep in 1 2 3; do
for wl in "narrow_pss728 --tile 500 --participants 3000" "narrow_pss19682 --tile 40 --participants 240"; do
  echo "== $wl (rep $rep)"
  for n in "$@"; do
    echo "$n : $(SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_$n.so fused --workload $wl)"
  done
done; done
# kernel durations of the LAST variant named (gen kernel against the follow-up kernel)
L=$PWD/sda_amd/lib/libsda_hip_${@: -1}.so
cd /tmp && export TMPDIR=/tmp
for wl in "narrow_pss728 --tile 500 --participants 2000" "narrow_pss19682 --tile 40 --participants 160"; do
  name=$(echo $wl | cut -d' ' -f1); rm -rf /tmp/prof_$name
  SDA_HIP_LIBRARY=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-additional --no-verify --workload $wl > /dev/null 2>&1
  echo "== kernel stats $name (${@: -1})"; for f in $(find /tmp/prof_$name -name "*kernel_stats.csv"); do head -6 "$f" | cut -c1-220; done
done
