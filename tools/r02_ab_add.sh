R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
echo "== additive"; bash tools/ab_libs_fused.sh gpurun_ab/libsda_cur.so gpurun_ab/libsda_addu2.so -- --workload additive --participants 40000 --tile 2000 2>&1
