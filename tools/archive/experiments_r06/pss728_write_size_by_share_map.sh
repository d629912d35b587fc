cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for map in sys tss; do
  rm -rf /tmp/pw_$map
  if [ $map = tss ]; then export SDA_BENCH_SHARE_MAP=tss; else unset SDA_BENCH_SHARE_MAP; fi
  timeout 300 rocprofv3 -i $R/tools/pmc_write.txt --kernel-trace --output-format csv -d /tmp/pw_$map -- python $R/bench.py --schedule serial --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-additional --workload narrow_pss728 --tile 500 --participants 1500 > /dev/null 2>&1
  python3 - $map <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob('/tmp/pw_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        if 'sda::' in row['Kernel_Name']:
            d[(row['Kernel_Name'].split('(')[0][:60], row['Counter_Name'])].append(float(row['Counter_Value']))
for k, v in d.items(): print(sys.argv[1], k, 'mean GB', sum(v)/len(v)*1024/1e9, 'n', len(v))
PY
done
