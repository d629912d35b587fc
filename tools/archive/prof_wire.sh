cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_wire; mkdir -p $O
ROWS=${ROWS:-16000} rocprofv3 -i $R/tools/pmc_sq.txt --kernel-trace --output-format csv -d $O/sq -- python $R/tools/bench_wire.py > $O/log.txt 2>&1
ROWS=${ROWS:-16000} rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_wire.py >> $O/log.txt 2>&1
cd $R && python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/prof_wire/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row['Kernel_Name'].split('(')[0][:60], row['Counter_Name']); acc[k][0] += float(row['Counter_Value']); acc[k][1] += 1
for n in sorted({k[0] for k in acc}):
    if 'varint' not in n and 'combine' not in n: continue
    print(n)
    print('   ' + '  '.join('%s=%.3g' % (c, v / cnt) for (kn, c), (v, cnt) in sorted(acc.items()) if kn == n))
for f in glob.glob('gpurun_out/prof_wire/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print('%-70s calls %s avg %.3f ms' % (row['Name'].split('(')[0][:70], row['Calls'], float(row['AverageNs']) / 1e6))
PY
