# A/B of two builds of the library on ONE box (clocks differ between boxes): narrow_pss728, serial and fused, two repetitions.
# usage (on the GPU box): bash tools/ab_ngemm.sh   - compares sda_amd/lib/libsda_hip_headng.so (a build with another ngemm_kernels.hip) with the in-tree library
mkdir -p gpurun_out
for rep in 1 2; do for v in headng cur; do for s in serial fused; do
  if [ $v = cur ]; then unset SDA_HIP_LIBRARY; else export SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_$v.so; fi
  timeout 300 python bench.py --workload narrow_pss728 --schedule $s --steps 4 --warmup 1 --participants 4000 --tile 500 --no-cpu-baseline --no-verify --no-additional --details gpurun_out/ab_tmp.details.json > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
  python - <<P
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$v $s rep$rep", round(d["value"]/1e9,2), "Gelem/s  ms/step", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "launch ms", r.get("avg_launch_ms"), r.get("both_roles_launch_ms"))
P
done; done; done
