# round-6 evidence (the round-5 script with the limb GEMM in its DEFAULT schedule - clerk waves inside the share-generation kernel): for every bench workload, at ONE tile size per workload and in the DEFAULT schedule (dual-role launch;
# the transform shape has none and runs serial): kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic (separate
# --pmc passes FETCH_SIZE / WRITE_SIZE, kernel-trace only) and the SQ counters that say which ceiling is active
# (tools/pmc_sq.txt, two more passes).   bash tools/profile_r06.sh [workloads...]  ->  gpurun_out/r06_final/<workload>/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_final; mkdir -p $O
# tiles = the ones bench.py runs in the driver form (its `roofline.traffic` is a measurement of exactly that tile or null)
declare -A TILE=( [packed]=2500 [additive]=2000 [packed26]=1250 [packed_ref]=1500 [packed26_ref]=1500 [packed_dim16m]=125 [packed_pss728]=500 [narrow_ref]=2000 [narrow26_ref]=2000 [narrow_pss728]=500 [narrow_pss19682]=40 )
declare -A SCHED=( [packed_pss728]="--schedule serial" )
# (name@tile: the same workload at another tile - packed@2000 is what `python bench.py` without flags runs, 50 steps of 2000)
WLS="${@:-packed packed@2000 additive packed26 packed_dim16m packed_pss728 narrow_ref narrow26_ref narrow_pss728 narrow_pss19682}"
cd /tmp && export TMPDIR=/tmp
for WT in $WLS; do
  W=${WT%@*}; T=${TILE[$W]}; D=$O/$W; [ "$WT" != "$W" ] && { T=${WT#*@}; D=$O/${W}_tile$T; }
  S=${SCHED[$W]}; mkdir -p $D
  A="--full-line --workload $W --tile $T --no-cpu-baseline --no-verify --no-additional $S"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python $R/bench.py $A --steps 10 --warmup 2 --participants $((10*T)) 2>$D/rocprof_stats.log | tail -1 > $D/bench_under_rocprof.json
  for c in fetch write sq; do
    timeout 600 rocprofv3 -i $R/tools/pmc_$c.txt --kernel-trace --output-format csv -d $D/pmc_$c -- python $R/bench.py $A --steps 4 --warmup 1 --participants $((4*T)) > /dev/null 2>$D/rocprof_$c.log
  done
  find $D/stats -name '*kernel_stats.csv' -exec cp {} $D/kernel_stats.csv \;
  python3 - "$D" <<'PY'
import csv, glob, sys, collections, json
D = sys.argv[1]
def rows_of(c):
    rows = []
    for f in glob.glob(D + '/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'sda::' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
    return rows
# The limb GEMM's share-generation kernel has ONE grid size for its share-generation-only launch (the first of a run) and for the
# launches that also carry the clerk waves' sum of the previous tile: keep the both-roles launches only - the k-th launch of the
# kernel is the same launch in every pass, and the FETCH pass tells the two kinds apart (the clerk waves read a whole tile of shares).
def launch_lists(c):
    per = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [(grid, value)] in dispatch order
    for r in rows_of(c):
        per[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append((r['Grid_Size'], float(r['Counter_Value'])))
    return per
fetch = launch_lists('fetch')
keep = {}
for k, cs in fetch.items():
    if 'packed_gen_ngemm_kernel' in k and 'FETCH_SIZE' in cs:
        vals = [v for _, v in cs['FETCH_SIZE']]
        keep[k] = [i for i, v in enumerate(vals) if v >= 0.5 * max(vals)] if max(vals) > 2.5 * min(vals) else list(range(len(vals)))
def means(per):
    out = collections.defaultdict(dict)
    for k, cs in per.items():
        for c, lst in cs.items():
            if k in keep:
                lst = [lst[i] for i in keep[k] if i < len(lst)]
            by_grid = collections.defaultdict(list)
            for g, v in lst:
                by_grid[g].append(v)
            for g, v in by_grid.items():
                out[(k, g)][c] = (sum(v) / len(v), len(v))
    return out
out = {}
for c in ('fetch', 'write'):
    for (k, g), cs in sorted(means(launch_lists(c)).items()):
        for name, (m, n) in cs.items():
            out['%s :: %s :: grid %s' % (k, name, g)] = {'mean_KiB': m, 'launches': n}
json.dump(out, open(D + '/pmc_hbm.json', 'w'), indent=1)
sq = collections.defaultdict(dict)
for (k, g), cs in sorted(means(launch_lists('sq')).items()):
    for name, (m, n) in cs.items():
        sq['%s :: grid %s' % (k, g)][name] = m
        sq['%s :: grid %s' % (k, g)]['launches'] = n
json.dump(sq, open(D + '/pmc_sq.json', 'w'), indent=1)
json.dump({k: v for k, v in keep.items()}, open(D + '/both_roles_launch_indices.json', 'w'))
PY
  rm -rf $D/stats $D/pmc_fetch $D/pmc_write $D/pmc_sq
  echo "== $W"; head -3 $D/kernel_stats.csv | cut -c1-200
done
