cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-additional --full-line --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; e=r['launch_ms_each'][1:-1]; print('pad %10s: %.1f Gelem/s even launches %.2f odd %.2f' % (os.environ.get('SDA_BENCH_ARENA_PAD','separate'), d['value']/1e9, sum(e[0::2])/len(e[0::2]), sum(e[1::2])/len(e[1::2])))"; }
for rep in 1 2 3; do
  unset SDA_BENCH_ARENA_PAD; run
  for pad in 0 4096 65536 1048576 2097152 16777216 268435456 1073741824; do export SDA_BENCH_ARENA_PAD=$pad; run; done
done
