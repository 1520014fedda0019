#!/bin/bash
# Measurement set of one round, run on the MI355X box:  gpurun --timeout 2400 -- 'bash profiles/measure.sh r2'
# then here:  python profiles/refresh.py gpurun_out/r2 r02
# (PMC passes carry --kernel-trace only through the counter collection itself: no --sys-trace / hip / hsa domains.)
set -u
tag=${1:-set}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
# 1. the headline line (BASELINE configs[1]) and the two other single-GPU configs, as the driver runs them
timeout 600 python bench.py > "$out/bench_full.json" 2> "$out/bench_full.err"
timeout 400 python bench.py --kind PARK --preset parkinglot --scans 2000 > "$out/bench_park.json" 2> "$out/bench_park.err"
timeout 400 python bench.py --kind OS128 --preset os128_fine --scans 1000 > "$out/bench_os128.json" 2> "$out/bench_os128.err"
cd /tmp
# 2. rocprofv3 kernel trace + stats of the same command at 1024 scans (the trace of 2761 scans x 3 steps is only bigger)
for cfg in "k64:--scans 1024" "park:--kind PARK --preset parkinglot --scans 1024" "os128:--kind OS128 --preset os128_fine --scans 512"; do
    name=${cfg%%:*}; args=${cfg#*:}
    rm -rf /tmp/prof_stats_$name
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$name -- python "$R/bench.py" $args --steps 2 --no-cpu --no-extras \
        > "$out/bench_under_rocprof_$name.json" 2> "$out/rocprof_stats_$name.err"
    f=$(find /tmp/prof_stats_$name -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$out/rocprofv3_kernel_stats_$name.csv"
done
# 3. HBM traffic PER WORKLOAD: FETCH_SIZE and WRITE_SIZE in separate passes (one step, no warm-up: one dispatch per kernel)
# (the bench line's own job sizes: the tracking chain's traffic per scan depends on how many walkers a job is cut into)
for cfg in "k64:2761:--scans 2761" "park:2000:--kind PARK --preset parkinglot --scans 2000" "os128:1000:--kind OS128 --preset os128_fine --scans 1000"; do
    name=${cfg%%:*}; rest=${cfg#*:}; nsc=${rest%%:*}; args=${rest#*:}
    for c in FETCH_SIZE WRITE_SIZE; do
        d=/tmp/prof_${c}_$name
        rm -rf $d
        timeout 900 rocprofv3 --pmc $c --kernel-include-regex scvod --output-format csv -d $d -- python "$R/bench.py" $args --steps 1 --warmup 0 --no-cpu --no-extras \
            > /dev/null 2> "$out/rocprof_${c}_$name.err"
        f=$(find $d -name '*counter_collection.csv' | head -1)
        [ -n "$f" ] && (head -1 "$f"; grep scvod "$f") > "$out/${c}_counter_collection_$name.csv"
    done
    echo $nsc > "$out/pmc_scans_$name.txt"
done
# 4. SQ counters (what bounds a kernel: issue vs wait, VALU vs LDS share, LDS bank conflicts), 256 scans
d=/tmp/prof_sq
rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-include-regex scvod --output-format csv -d $d -- python "$R/bench.py" --scans 256 --steps 1 --warmup 0 --no-cpu --no-extras \
    > /dev/null 2> "$out/rocprof_sq.err"
f=$(find $d -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && (head -1 "$f"; grep scvod "$f") > "$out/SQ_counter_collection.csv"
# 5. phase clocks inside k_cc_scan and the large sort tiers (development build of the library: make -C dr-using-scv-od_amd/csrc prof)
if [ -f "$R/dr-using-scv-od_amd/csrc/libscvod_prof.so" ]; then
    (timeout 200 python "$R/tools/kernel_phases.py" --scans 512; timeout 200 python "$R/tools/kernel_phases.py" --kind OS128 --preset os128_fine --scans 128;
     timeout 200 python "$R/tools/kernel_phases.py" --kind PARK --preset parkinglot --scans 512) > "$out/kernel_phases.txt" 2> "$out/kernel_phases.err"
fi
ls -la "$out"
