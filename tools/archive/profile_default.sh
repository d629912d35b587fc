# round profile of the default bench (run on the GPU box through gpurun): tests, bench line, kernel stats, HBM PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-additional"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $BARGS > $O/bench_under_rocprof.json 2>$O/rocprof_stats.log
for c in fetch write; do
  timeout 600 rocprofv3 -i $R/tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-additional > /dev/null 2>$O/rocprof_$c.log
done
cd $R && python - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    print(open(f).read())
out = {}
for c in ('fetch', 'write'):
    d = collections.defaultdict(list)
    for f in glob.glob(O + '/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for row in csv.DictReader(open(f)):
            d[(row['Kernel_Name'].split('(')[0], row['Counter_Name'], row['Grid_Size'])].append(float(row['Counter_Value']))
    for k, v in sorted(d.items()):
        out['%s :: %s :: grid %s' % k] = {'mean': sum(v) / len(v), 'launches': len(v)}
json.dump(out, open(O + '/pmc_hbm.json', 'w'), indent=1)
for k, v in out.items():
    if 'sda::' in k: print(k, v)
print(open(O + '/bench_default.json').read()[:1500])
PY
