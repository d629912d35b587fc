R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k transform_share 2>&1 | grep -E "passed|failed"
for t in 256 512 1024 128; do for i in 1 2; do SDA_FFT_THREADS=$t python bench.py --workload packed_pss728 --steps 4 --warmup 1 --participants 2000 --schedule serial --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads $t', d['kernels']['share_gen']['avg_ms'], d['value']/1e9)"; done; done
