#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03
# Per workload (C2 = BASELINE configs[1]: 128x128, batch 32, Euler-Maruyama;  C4 = configs[3]: 256x256, batch 16, predictor-corrector):
# 1. bench.py autotune -> tile table saved
# 2. rocprofv3 --kernel-trace --stats of the same command (same tile table)          -> <tag>_<wl>_kernel_stats.csv, _step_breakdown.txt
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) -> <tag>_pmc_traffic_<wl>.json (+ source hash)
# 4. bench.py: the JSON line (roofline.traffic from 3) and the per-convolution HIP-event timings
# then the C3 training step (kernel trace -> <tag>_train_step_breakdown.txt, bench line).
# Everything lands in gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -eo pipefail
TAG=${1:-r03}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT

one_workload () {   # name  workload-key  steps  extra bench args...
    local NAME=$1 WL=$2 STEPS=$3; shift 3
    local T=$OUT/${TAG}_${NAME}_tiles.txt
    rm -f $T
    # autotune once; every later pass replays this tile table
    python3 bench.py --steps 20 --warmup 10 --tune-cache $T --no-cpu-baseline --no-secondary "$@" > /dev/null 2> $OUT/${NAME}_tune.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
        python3 bench.py --steps 40 --warmup 5 --tune-cache $T --no-cpu-baseline --no-secondary "$@" > $OUT/stats.log 2>&1
    python3 tools/step_breakdown.py $OUT/stats/bench_kernel_trace.csv > $OUT/${TAG}_${NAME}_step_breakdown.txt
    cp $OUT/stats/bench_kernel_stats.csv $OUT/${TAG}_${NAME}_kernel_stats.csv
    rm -rf $OUT/stats
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- \
        python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --tune-cache $T "$@" > $OUT/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- \
        python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --tune-cache $T "$@" > $OUT/pmc_write.log 2>&1
    python3 tools/pmc_traffic.py $OUT/pmc_fetch/fetch_counter_collection.csv $OUT/pmc_write/write_counter_collection.csv \
        $OUT/${TAG}_pmc_traffic_${WL}.json $WL > $OUT/${TAG}_pmc_traffic_${WL}.txt
    rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/*.log
    # the bench line last: it reads the traffic file measured above (same kernel sources, same tile table) from profiles/
    cp $OUT/${TAG}_pmc_traffic_${WL}.json profiles/
    python3 bench.py --steps $STEPS --warmup 10 --tune-cache $T --profile-csv $OUT/${TAG}_${NAME}_conv_launches_hip_events.csv "$@" \
        > $OUT/${TAG}_${NAME}_bench_line.json 2> $OUT/${NAME}_bench.err
    echo "== $NAME"; cut -c1-260 $OUT/${TAG}_${NAME}_bench_line.json; head -3 $OUT/${TAG}_${NAME}_step_breakdown.txt
}

one_workload c2 b32_128_em 200
one_workload c4 b16_256_pc 60 --size 256 --batch 16 --sampler pc

# C3 training step: bench line + per-kernel breakdown of one optimizer step
python3 bench.py --mode train --steps 30 --warmup 5 > $OUT/${TAG}_train_bench_line.json 2> $OUT/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tstats -o train -- \
    python3 bench.py --mode train --steps 12 --warmup 4 > $OUT/tstats.log 2>&1
python3 tools/train_breakdown.py $OUT/tstats/train_kernel_trace.csv > $OUT/${TAG}_train_step_breakdown.txt || true
rm -rf $OUT/tstats $OUT/*.log
echo "== train"; cut -c1-260 $OUT/${TAG}_train_bench_line.json; head -12 $OUT/${TAG}_train_step_breakdown.txt
