# round-2 evidence: kernel stats (rocprofv3 --kernel-trace --stats) and HBM traffic (separate --pmc passes, kernel-trace only)
# for every bench workload, at ONE tile size per workload; plus the driver-form default line.   bash tools/profile_r02.sh [workloads...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_final; mkdir -p $O
declare -A TILE=( [packed]=2500 [additive]=2000 [packed26]=1500 [packed_ref]=1500 [packed26_ref]=1500 [packed_dim16m]=125 [packed_pss728]=500 )
declare -A SCHED=( [packed_pss728]="--schedule serial" )
WLS="${@:-packed additive packed26 packed_ref packed26_ref packed_dim16m packed_pss728}"
cd $R && timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_driver_form.err | tail -1 > $O/bench_driver_form.json
cd /tmp && export TMPDIR=/tmp
for W in $WLS; do
  T=${TILE[$W]}; S=${SCHED[$W]}; D=$O/$W; mkdir -p $D
  A="--workload $W --tile $T --no-cpu-baseline --no-verify --no-additional $S"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python $R/bench.py $A --steps 10 --warmup 2 --participants $((10*T)) 2>$D/rocprof_stats.log | tail -1 > $D/bench_under_rocprof.json
  for c in fetch write; do
    timeout 600 rocprofv3 -i $R/tools/pmc_$c.txt --kernel-trace --output-format csv -d $D/pmc_$c -- python $R/bench.py $A --steps 3 --warmup 1 --participants $((3*T)) > /dev/null 2>$D/rocprof_$c.log
  done
  find $D/stats -name '*kernel_stats.csv' -exec cp {} $D/kernel_stats.csv \;
  python3 - "$D" <<'PY'
import csv, glob, sys, collections, json
D = sys.argv[1]
out = {}
for c in ('fetch', 'write'):
    d = collections.defaultdict(list)
    for f in glob.glob(D + '/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'sda::' in row['Kernel_Name']:
                d[(row['Kernel_Name'].split('(')[0], row['Counter_Name'], row['Grid_Size'])].append(float(row['Counter_Value']))
    for k, v in sorted(d.items()):
        out['%s :: %s :: grid %s' % k] = {'mean_KiB': sum(v) / len(v), 'launches': len(v)}
json.dump(out, open(D + '/pmc_hbm.json', 'w'), indent=1)
PY
  rm -rf $D/stats $D/pmc_fetch $D/pmc_write
  echo "== $W"; head -3 $D/kernel_stats.csv | cut -c1-200
done
