#!/bin/bash
# The A/B variant of the library: identical sources, -DSDA_AB_KNOBS, so that an UNSET knob falls back to the environment
# variable of the same name (include/sda_hip_debug.h) - for shell-driven A/B runs of tools that do not pass knobs themselves.
# Written to sda_amd/lib/libsda_hip_ab.so; never loaded by the package (sda_amd/capi.py loads libsda_hip.so) unless
# SDA_HIP_LIBRARY points at it.  __graft_entry__.build() never builds this.
set -e
cd "$(dirname "$0")/.."
OUT=sda_amd/lib/ab_obj; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSDA_AB_KNOBS"
for f in sda_kernels.hip varint_kernels.hip wire_kernels.hip sealedbox_kernels.hip fft_kernels.hip ngemm_kernels.hip signed_kernels.hip narrow_kernels.hip sda_capi.cpp sda_comm.cpp sda_wire.cpp sda_sealedbox.cpp; do
  [ -f sda_amd/csrc/$f ] || continue
  (cd /tmp && /opt/rocm/bin/hipcc $FLAGS -c $OLDPWD/sda_amd/csrc/$f -o $OLDPWD/$OUT/$f.o) &
done
wait
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $OLDPWD/$OUT/*.o -ldl -o $OLDPWD/sda_amd/lib/libsda_hip_ab.so)
echo built sda_amd/lib/libsda_hip_ab.so
