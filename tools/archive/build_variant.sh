# usage: bash tools/build_variant.sh <name> [extra hipcc flags]  ->  gpurun_ab/libsda_<name>.so  (A/B builds; travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift; mkdir -p $R/gpurun_ab
SRC=$(python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; print(' '.join('$R/sda_amd/csrc/'+f for f in g.SOURCES))")
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,-rpath,/opt/rocm/lib "$@" $SRC -ldl -o $R/gpurun_ab/libsda_$N.so && echo built $R/gpurun_ab/libsda_$N.so
