#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01
# 1. bench.py (autotune -> tile table saved), the JSON line and the per-convolution HIP-event timings
# 2. rocprofv3 --kernel-trace --stats of the same command (same tile table)      -> <tag>_bench_kernel_stats.csv
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) -> <tag>_pmc_traffic.json
# Everything lands in gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -eo pipefail
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
T=$OUT/${TAG}_tiles.txt
rm -f $T
python3 bench.py --steps 200 --warmup 20 --tune-cache $T --profile-csv $OUT/${TAG}_conv_launches_hip_events.csv \
    > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
    python3 bench.py --steps 200 --warmup 20 --tune-cache $T --no-cpu-baseline > $OUT/stats.log 2>&1
python3 tools/step_breakdown.py $OUT/stats/bench_kernel_trace.csv > $OUT/${TAG}_sde_step_breakdown.txt
cp $OUT/stats/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/stats
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- \
    python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --tune-cache $T > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- \
    python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --tune-cache $T > $OUT/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch/fetch_counter_collection.csv $OUT/pmc_write/write_counter_collection.csv \
    $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/*.log
cat $OUT/${TAG}_bench_line.json | cut -c1-300
