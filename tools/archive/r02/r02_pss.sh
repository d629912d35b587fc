R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_pss; mkdir -p $O
cd $R
timeout 900 python bench.py --workload packed_pss728 --steps 10 --warmup 1 --schedule serial --no-cpu-baseline > $O/bench_pss728_fft.json 2>$O/err1.txt; tail -2 $O/err1.txt; python -c "
import json;d=json.loads(open('$O/bench_pss728_fft.json').read().splitlines()[-1]);print('FFT', d['value']/1e9,'Gelem/s', d['kernels'], d['verified_reconstruct_equals_sum'], d['path_roofline'])"
SDA_FORCE_GENERIC=1 timeout 900 python bench.py --workload packed_pss728 --steps 2 --warmup 0 --participants 40 --schedule serial --no-cpu-baseline > $O/bench_pss728_generic.json 2>$O/err2.txt; tail -2 $O/err2.txt; python -c "
import json;d=json.loads(open('$O/bench_pss728_generic.json').read().splitlines()[-1]);print('GENERIC', d['value']/1e9,'Gelem/s', d['kernels'], d['verified_reconstruct_equals_sum'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --workload packed_pss728 --steps 4 --warmup 1 --participants 2000 --schedule serial --no-cpu-baseline --no-verify > /dev/null 2>$O/rocprof.log
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pss728.csv \; ; head -5 $O/kernel_stats_pss728.csv
