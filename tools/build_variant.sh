#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  ->  variants/libskx_<name>.so (select with SKX_LIB_PATH)
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/variants; mkdir -p $O/obj_$N
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I $R/include -I $R/sketchy_amd/csrc -Wall -Wno-unused-result"
hipcc $F "$@" -c $R/sketchy_amd/csrc/skx_kernels.hip -o $O/obj_$N/k.o &
hipcc $F "$@" -c $R/sketchy_amd/csrc/skx_capi.hip -o $O/obj_$N/c.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $O/obj_$N/k.o $O/obj_$N/c.o -o $O/libskx_$N.so -ldl
rm -rf $O/obj_$N; echo $O/libskx_$N.so
