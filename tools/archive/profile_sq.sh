# SQ counters (VALU busy / wait breakdown) of the share-gen kernels, serial schedule so that each kernel stands alone:
# bash tools/profile_sq.sh <workload> <tile>   ->  gpurun_out/r02_sq/<workload>/pmc_sq.json
R=${GRAFT_REPO_ROOT:-/root/repo}; W=$1; T=$2; O=$R/gpurun_out/r02_sq/$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 -i $R/tools/pmc_sq.txt --kernel-trace --output-format csv -d $O/raw -- python $R/bench.py --workload $W --tile $T --participants $((3*T)) --steps 3 --warmup 1 --schedule serial --no-cpu-baseline --no-verify --no-additional > /dev/null 2>$O/rocprof.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(O + '/raw/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'sda::' in row['Kernel_Name'] and ('gen' in row['Kernel_Name'] or 'combine_update' in row['Kernel_Name']):
            d[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
out = collections.defaultdict(dict)
for (k, c), v in d.items():
    out[k][c] = sum(v) / len(v)
for k, c in out.items():
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        c['_valu_active_share_of_wave_cycles'] = c.get('SQ_ACTIVE_INST_VALU', 0) / wc
        c['_wait_inst_any_share'] = c.get('SQ_WAIT_INST_ANY', 0) / wc
        c['_wait_any_share'] = c.get('SQ_WAIT_ANY', 0) / wc
        c['_valu_insts_per_wave'] = c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1)
json.dump(out, open(O + '/pmc_sq.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $O/raw
