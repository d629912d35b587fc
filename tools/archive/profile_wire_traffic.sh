# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the wire-format kernels at ROWS rows
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_wire_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in fetch write; do
  ROWS=${ROWS:-16000} timeout 600 rocprofv3 -i $R/tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/tools/bench_wire.py > /dev/null 2>$O/rocprof_$c.log
done
cd $R && python - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
for c in ('fetch', 'write'):
    d = collections.defaultdict(list)
    for f in glob.glob(O + '/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for row in csv.DictReader(open(f)):
            d[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    for k, v in sorted(d.items()):
        if 'varint' in k[0] or 'combine_update' in k[0]:
            out['%s :: %s' % k] = {'mean_KiB': sum(v) / len(v), 'launches': len(v)}
json.dump(out, open(O + '/pmc_wire.json', 'w'), indent=1)
for k, v in out.items(): print('%-70s %.2f GB x%d' % (k, v['mean_KiB'] * 1024 / 1e9, v['launches']))
PY
