# usage: bash tools/ab_libs_fused.sh libA.so libB.so ... -- [bench args]   interleaved A/B of several builds, default schedule
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
run() { SDA_HIP_LIBRARY=$1 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-verify --no-additional "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-26s %.1f Gelem/s launch %.3f ms (both roles %.3f)' % (sys.argv[1], d['value']/1e9, r['avg_launch_ms'], r['both_roles_launch_ms']))" "$(basename $1)"; }
for i in 1 2 3; do for l in "${LIBS[@]}"; do run $l "$@"; done; done
