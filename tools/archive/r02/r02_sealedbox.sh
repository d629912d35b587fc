R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_sbox; mkdir -p $O
cd $R && timeout 600 python tools/bench_sealedbox.py > $O/sealedbox_bench.json 2>$O/err.txt; cat $O/sealedbox_bench.json; tail -3 $O/err.txt
cd /tmp && export TMPDIR=/tmp
REPS=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_sealedbox.py > /dev/null 2>$O/rocprof.log
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_sealedbox.csv \; ; head -12 $O/kernel_stats_sealedbox.csv
