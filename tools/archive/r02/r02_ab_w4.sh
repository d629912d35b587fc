R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for wl in packed26; do echo "== $wl"; bash tools/ab_libs_fused.sh gpurun_ab/libsda_g5.so gpurun_ab/libsda_g5w4.so -- --workload $wl --participants 30000 --tile 1500 2>&1; done
