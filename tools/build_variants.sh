#!/bin/bash
# Build variants of ONE translation unit here (no GPU needed) and link each into py_neuromodulation_amd/libnmx_v<k>.so next
# to the product library; the GPU-box script copies a variant over libnmx.so in its scratch copy (tools/exp_lib_variants.sh).
#   tools/build_variants.sh nmx_specmm.hip nmx_specmm.o "" "-DNMX_SMM_DEBUG_NOCOMP" ...
set -e
cd "$(dirname "$0")/../py_neuromodulation_amd/csrc"
SRC=$1; OBJ=$2; shift 2
k=0
for flags in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $SRC -o _build/v${k}_$OBJ &&
    objs=$(ls _build/*.o | grep -v "/v[0-9]*_" | grep -v "/$OBJ") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs _build/v${k}_$OBJ -o ../libnmx_v${k}.so &&
    echo "v$k: $flags" ) &
  k=$((k+1))
done
wait
