R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -4
for wl in packed26 packed26_ref packed_ref; do echo "== $wl"; bash tools/ab_libs_fused.sh gpurun_ab/libsda_lbw.so gpurun_ab/libsda_g5.so -- --workload $wl --participants 30000 --tile 1500 2>&1; done
