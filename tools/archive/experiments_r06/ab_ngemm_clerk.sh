#!/bin/bash
# limb GEMM, dual-role launch: clerk WAVES inside the share-generation workgroups (default) against clerk WORKGROUPS in the grid
# (rounds 4 - 5; knob SDA_NGEMM_CLERK_WG=1), interleaved; then rocprofv3 kernel stats of the default form (how much the follow-up
# kernel had left to do)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-additional --full-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f Gelem/s frac %.3f launch %.3f ms first(gen only) %.3f last(clerk only) %.3f verified %s' % (d['value']/1e9, r['frac'], r.get('both_roles_launch_ms') or 0, r.get('first_launch_ms_share_gen_only') or 0, r.get('last_launch_ms_clerk_sum_only') or 0, d['verified_reconstruct_equals_sum']))"; }
for rep in 1 2; do
for wl in "narrow_pss728 --tile 500 --participants 3000" "narrow_pss19682 --tile 40 --participants 240"; do
  echo "== $wl (rep $rep)"
  echo "clerk waves      : $(fused --workload $wl)"
  echo "clerk workgroups : $(SDA_NGEMM_CLERK_WG=1 fused --workload $wl)"
done; done
cd /tmp && export TMPDIR=/tmp
for wl in "narrow_pss728 --tile 500 --participants 2000" "narrow_pss19682 --tile 40 --participants 160"; do
  name=$(echo $wl | cut -d' ' -f1)
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-additional --no-verify --workload $wl > /dev/null 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats $name"; head -8 "$f" | cut -c1-200
done
