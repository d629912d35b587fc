# HBM PMC passes + kernel stats for one bench workload: bash tools/profile_workload.sh <name> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; N=$1; shift; O=$R/gpurun_out/prof_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 10 --warmup 2 --full-line --no-cpu-baseline --no-verify --no-additional "$@" > $O/bench_under_rocprof.json 2>$O/rocprof_stats.log
for c in fetch write; do
  timeout 600 rocprofv3 -i $R/tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-additional "$@" > /dev/null 2>$O/rocprof_$c.log
done
cd $R && python - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:1200])
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
PY
