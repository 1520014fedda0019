#!/bin/bash
# development: bench.py on several builds of the library (csrc/libscvod_<name>.so) on ONE box: tools/ab_bench.sh <tag> <name>... [-- bench args]
tag=$1; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p gpurun_out/$tag
for v in "${names[@]}"; do
    lib=${v%%-*}; bal=1   # name[-nobal]: libscvod_<name>.so, -nobal = equal-length chain segments (SCVOD_CHAIN_BALANCE=0)
    case "$v" in *-nobal*) bal=0;; esac
    SCVOD_CHAIN_BALANCE=$bal SCVOD_LIB=dr-using-scv-od_amd/csrc/libscvod_$lib.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu --no-extras "$@" > gpurun_out/$tag/bench_$v.json 2> gpurun_out/$tag/bench_$v.err
done
python tools/ab_report.py gpurun_out/$tag "${names[@]}"
