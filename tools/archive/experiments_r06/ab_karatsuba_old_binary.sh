cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line --workload packed26 --tile 1250 --participants 100000 --steps 20 --warmup 2 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s %.1f Gelem/s frac %.4f both-roles %.3f ms verified %s' % (sys.argv[1], d['value']/1e9, r['frac'], r['both_roles_launch_ms'], d['verified_reconstruct_equals_sum']))" "$1"; }
for rep in 1 2 3; do
  SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_oldk.so run "old kernels (round 5 form)"
  SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_test.so run "new, karatsuba"
  SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_test.so SDA_NO_KARATSUBA=1 run "new, plain wide group"
done
