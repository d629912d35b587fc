#!/bin/bash
# kernel durations + SQ counters of the narrow limb GEMM on tss's PSS_155_728_100 over 746497 (serial schedule: the share
# generation kernel alone, then the clerk sum)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_ngemm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--workload narrow_pss728 --tile 500 --participants 2000 --schedule serial --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-additional"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $A > $O/bench_under_rocprof.json 2>$O/rocprof_stats.log
cat > /tmp/pmc_ng.txt <<'PMC'
pmc: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc: GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
PMC
timeout 300 rocprofv3 -i /tmp/pmc_ng.txt --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/bench.py $A > /dev/null 2>$O/rocprof_sq.log
cd $R && python - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:1500])
d = collections.defaultdict(list)
for f in glob.glob(O + '/pmc_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        d[(row['Kernel_Name'].split('(')[0][:60], row['Counter_Name'])].append(float(row['Counter_Value']))
out = {'%s :: %s' % k: sum(v) / len(v) for k, v in sorted(d.items())}
json.dump(out, open(O + '/pmc_sq.json', 'w'), indent=1)
for k, v in out.items():
    if 'ngemm' in k or 'fft' in k: print(k, v)
PY
tail -3 $O/rocprof_sq.log
