# other sizes through the default (dual-role) schedule: config-5 dimension, odd dimensions, larger tiles
run() { python bench.py --no-cpu-baseline --no-additional "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-56s %.1f Gelem/s step %.2f ms %s %.3f of peak ok=%s' % (sys.argv[1], d['value']/1e9, d['ms_per_step'], r['kernel'], r['frac'], d['verified_reconstruct_equals_sum']))" "$*"; }
run --workload packed --dim 16777216 --tile 128 --steps 10
run --workload packed --dim 16777216 --tile 256 --steps 5
run --workload packed --dim 1048576 --tile 4000 --steps 5
run --workload packed --dim 1000003 --tile 777 --steps 5
run --workload additive --dim 1000003 --tile 777 --steps 5
run --workload packed26 --dim 16777216 --tile 96 --steps 3
run --workload packed --dim 1000003 --tile 777 --steps 5 --row-align 1
