#!/bin/bash
# development: csrc/libscvod_<name>.so = the library with ONE unit rebuilt under extra -D flags: tools/ab_build.sh <name> <unit> "<flags>"
name=$1; unit=$2; flags=$3
cd "$(dirname "$0")/../dr-using-scv-od_amd/csrc" || exit 1
make -s libscvod.so || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-value $flags -c -o obj/${unit}.${name}.o ${unit}.hip || exit 1
objs=""
for u in scvod_kernels scvod_lastname scvod_track scvod_chain scvod_map scvod_capi; do
    if [ "$u" = "$unit" ]; then objs="$objs obj/${unit}.${name}.o"; else objs="$objs obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libscvod_${name}.so $objs
