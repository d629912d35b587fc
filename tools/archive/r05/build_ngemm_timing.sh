#!/bin/bash
# The phase-timing variant of the library (ngemm_kernels.hip with -DNG_TIMING: every limb-GEMM workgroup overwrites the first
# six shares of its first batch column with s_memtime cycle counts - staging, row tiles, and the staging passes of wave 0).
# For tools/dbg_ngemm_timing.py only; written to sda_amd/lib/libsda_hip_timing.so, never loaded by the package unless
# SDA_HIP_LIBRARY points at it; __graft_entry__.build() never builds it.   usage: tools/build_ngemm_timing.sh && \
#   SDA_HIP_LIBRARY=$PWD/sda_amd/lib/libsda_hip_timing.so python tools/dbg_ngemm_timing.py 256
set -e
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build()"
OBJ=sda_amd/lib/obj
# NG_DEFINE=NG_PHASES: only the two phases (staging / row tiles) - the per-pass timers of NG_TIMING cost registers, and the row loop
# of an instrumented build is only comparable with the shipped one when it is as free of scratch reloads (tests/test_ngemm_isa.py)
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -D${NG_DEFINE:-NG_TIMING} -c $OLDPWD/sda_amd/csrc/ngemm_kernels.hip -o $OLDPWD/$OBJ/ngemm_timing.o)
SRC=$(python3 -c "import __graft_entry__ as g; print(' '.join('$OBJ/'+f+'.o' for f in g.SOURCES if f != 'ngemm_kernels.hip'))")
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $(for f in $SRC; do echo $OLDPWD/$f; done) $OLDPWD/$OBJ/ngemm_timing.o -ldl -o $OLDPWD/sda_amd/lib/libsda_hip_timing.so)
echo built sda_amd/lib/libsda_hip_timing.so
