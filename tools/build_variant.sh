#!/bin/bash
# Build a variant of the library with extra compiler flags for ff_gemm.hip only: build_ub/lib_<name>.so (use with FF_HIP_LIB=...).
# usage: tools/build_variant.sh <name> <flags...>      (run here, in the build container; the other objects are the in-tree ones)
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
python -m faceformer_amd.hip.build > /dev/null
mkdir -p build_ub/var_$name
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifaceformer_amd/csrc -Wall -Wno-unused-function "$@" -c faceformer_amd/csrc/ff_gemm.hip -o build_ub/var_$name/ff_gemm.o
objs=$(ls faceformer_amd/hip/build/*.o | grep -v ff_gemm.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o build_ub/lib_$name.so $objs build_ub/var_$name/ff_gemm.o
echo build_ub/lib_$name.so
