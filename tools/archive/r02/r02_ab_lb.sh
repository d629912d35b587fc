R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for wl in packed_ref packed26_ref packed26; do echo "== $wl"; bash tools/ab_libs_fused.sh gpurun_ab/libsda_base.so gpurun_ab/libsda_lbw.so -- --workload $wl --participants 30000 --tile 1500 2>&1 | sed -e 's/steps 20//' ; done
