cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_rows; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_rows.py > $O/log.txt 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_rows/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print('%-70s calls %s avg %.3f ms total %.1f ms' % (row['Name'].split('(')[0][:70], row['Calls'], float(row['AverageNs']) / 1e6, float(row['TotalDurationNs'])/1e6))
PY
