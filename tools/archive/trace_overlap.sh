# start / end timestamps of the kernels of a short pss728 run (are the transform kernel and the clerk sum concurrent?)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03_fft; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --workload packed_pss728 --steps 3 --warmup 1 --participants 1500 --no-cpu-baseline --no-verify --no-additional > /dev/null 2>$O/trace.log
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = []
for f in glob.glob(O + '/trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sda::', '')[:40]
    print('%-40s queue %s  start %9.3f ms  end %9.3f ms  dur %7.3f' % (n, r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
rm -rf $O/trace
