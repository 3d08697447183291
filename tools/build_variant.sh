#!/bin/bash
# Build a variant of the library with extra compiler flags for ONE source (FF_VARIANT_SRC, default ff_gemm): build_ub/lib_<name>.so
# (use with FF_HIP_LIB=...).
# usage: [FF_VARIANT_SRC=ff_attention] tools/build_variant.sh <name> <flags...>   (run here, in the build container; the other
#        objects are the in-tree ones)
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
python -m faceformer_amd.hip.build > /dev/null
mkdir -p build_ub/var_$name
src=${FF_VARIANT_SRC:-ff_gemm}
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifaceformer_amd/csrc -Wall -Wno-unused-function "$@" -c faceformer_amd/csrc/$src.hip -o build_ub/var_$name/$src.o
objs=$(ls faceformer_amd/hip/build/*.o | grep -v "/$src.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o build_ub/lib_$name.so $objs build_ub/var_$name/$src.o
echo build_ub/lib_$name.so
