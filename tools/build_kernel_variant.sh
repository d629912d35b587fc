#!/bin/bash
# A/B variant of the library that differs in ONE kernel file only: every other object is the in-tree build's (sda_amd/lib/obj, with
# the test hooks), FILE is recompiled with the given flags.   bash tools/build_kernel_variant.sh FILE.hip NAME [-DFLAG ...]
# -> sda_amd/lib/libsda_hip_NAME.so; load it with SDA_HIP_LIBRARY (never loaded otherwise; __graft_entry__.build() never builds it)
set -e
cd "$(dirname "$0")/.."
FILE=$1; NAME=$2; shift; shift
OBJ=sda_amd/lib/obj
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $OLDPWD/sda_amd/csrc/$FILE -o $OLDPWD/$OBJ/variant_$NAME.o)
OBJS=$(ls $OBJ/*.hip.o $OBJ/*.cpp.o | grep -v "$FILE.o" | grep -v "sda_capi.cpp.o")
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $(for o in $OBJS $OBJ/sda_capi.cpp.hooks.o $OBJ/variant_$NAME.o; do echo $OLDPWD/$o; done) -ldl -o $OLDPWD/sda_amd/lib/libsda_hip_$NAME.so)
echo built sda_amd/lib/libsda_hip_$NAME.so
