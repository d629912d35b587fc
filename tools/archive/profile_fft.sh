# SQ / LDS counters of the transform kernel alone (serial schedule):  bash tools/profile_fft.sh [tag]  ->  gpurun_out/r03_fft/<tag>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-run}; O=$R/gpurun_out/r03_fft; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 -i $R/tools/pmc_lds.txt --kernel-trace --output-format csv -d $O/raw_$TAG -- python $R/bench.py --workload packed_pss728 --schedule serial --steps 2 --warmup 1 --participants 1000 --no-cpu-baseline --no-verify --no-additional > /dev/null 2>$O/rocprof_$TAG.log
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys, collections, json
O, TAG = sys.argv[1:3]
d = collections.defaultdict(list)
for f in glob.glob(O + '/raw_%s/**/*counter_collection.csv' % TAG, recursive=True):
    for row in csv.DictReader(open(f)):
        if 'fft' in row['Kernel_Name']:
            d[row['Counter_Name']].append(float(row['Counter_Value']))
c = {k: sum(v) / len(v) for k, v in d.items()}
elems = 500 * 1048576
cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
c['_valu_wave_instr_per_element'] = c.get('SQ_INSTS_VALU', 0) / elems
c['_lds_wave_instr_per_element'] = c.get('SQ_INSTS_LDS', 0) / elems
c['_simd_cycles_per_valu_instr'] = cyc * 1024 / max(c.get('SQ_INSTS_VALU', 1), 1)
c['_valu_busy'] = c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / max(cyc, 1)
c['_lds_bank_conflict_share_of_lds_active'] = c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)
c['_lds_idx_active_share_of_gpu_cycles_per_cu'] = c.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / max(cyc, 1)
wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'):
    c['_share_of_wave_cycles_' + k] = c.get(k, 0) / wc
json.dump(c, open(O + '/%s.json' % TAG, 'w'), indent=1)
print(json.dumps(c, indent=1))
PY
rm -rf $O/raw_$TAG
