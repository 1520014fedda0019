#!/bin/bash
# Measurement set of one round, run on the MI355X box:  gpurun -- 'bash profiles/measure.sh r1d'
# then here:  python profiles/refresh.py gpurun_out/r1d r01
# (PMC passes carry --kernel-trace only through the counter collection itself: no --sys-trace / hip / hsa domains.)
set -u
tag=${1:-set}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench_full.json" 2> "$out/bench_full.err"
cd /tmp
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$R/bench.py" --scans 1024 --steps 2 --no-cpu --no-cpu-all --no-extras \
    > "$out/bench_under_rocprof.json" 2> "$out/rocprof_stats.err"
f=$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/rocprofv3_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/prof_$c
    rm -rf $d
    rocprofv3 --pmc $c --kernel-include-regex scvod --output-format csv -d $d -- python "$R/bench.py" --scans 512 --steps 1 --warmup 0 --no-cpu --no-cpu-all --no-extras \
        > /dev/null 2> "$out/rocprof_$c.err"
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && (head -1 "$f"; grep scvod "$f") > "$out/${c}_counter_collection.csv"
done
ls -la "$out"
