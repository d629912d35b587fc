#!/bin/bash
# narrow-modulus kernels against the 62-bit kernels on the SAME small primes (SDA_NO_NARROW=1 handed to the test-only knob by bench.py)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
fused() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-additional "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.1f Gelem/s frac %.3f launch %.3f ms verified %s kernel %s' % (d['value']/1e9, r['frac'], r['both_roles_launch_ms'] or r['avg_launch_ms'], d['verified_reconstruct_equals_sum'], r['kernel']))"; }
for w in ${WL:-narrow_ref narrow26_ref narrow_pss728}; do
  T=1500; [ $w = narrow_pss728 ] && T=500
  for i in 1 2; do
    echo "$w narrow: $(fused --workload $w --tile $T --participants $((8*T)))"
    echo "$w wide  : $(SDA_NO_NARROW=1 fused --workload $w --tile $T --participants $((8*T)))"
  done
done
for g in 8 16; do for th in 512 1024; do
  echo "narrow_pss728 G=$g threads=$th: $(SDA_FFT_G=$g SDA_FFT_THREADS=$th fused --workload narrow_pss728 --tile 500 --participants 4000)"
done; done
