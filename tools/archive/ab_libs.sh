# usage: bash tools/ab_libs.sh libA.so libB.so ... -- [bench args]   interleaved A/B of several builds
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
run() { SDA_HIP_LIBRARY=$1 python bench.py --schedule serial --steps 20 --warmup 2 --no-cpu-baseline --no-verify --no-additional "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-26s %.1f Gelem/s gen %.3f ms comb %.3f ms' % (sys.argv[1], d['value']/1e9, k['share_gen']['avg_ms'], k['clerk_sum']['avg_ms']))" "$(basename $1)"; }
for i in 1 2 3; do for l in "${LIBS[@]}"; do run $l "$@"; done; done
