#!/bin/bash
# Build a variant of the WHOLE library (every source compiled with the extra flags): build_ub/lib_<name>.so (use with FF_HIP_LIB=...).
# usage: tools/build_all_variant.sh <name> <flags...>     (run here, in the build container)
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_ub/all_$name
objs=""
for src in ff_rowops ff_gemm ff_gemm_x3 ff_attention ff_attention_x2h ff_pointer ff_engine; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifaceformer_amd/csrc -Wall -Wno-unused-function "$@" \
        -c faceformer_amd/csrc/$src.hip -o build_ub/all_$name/$src.o &
  objs="$objs build_ub/all_$name/$src.o"
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o build_ub/lib_$name.so $objs
echo build_ub/lib_$name.so
