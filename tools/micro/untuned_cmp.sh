#!/bin/bash
# static tile choice (engine.hip pick_tile) against the autotuned plan: ms per SDE step, EM sampler unless noted
# usage: bash tools/micro/untuned_cmp.sh [tuned]   -> gpurun_out/untuned_cmp.txt
out=gpurun_out/untuned_cmp.txt; mkdir -p gpurun_out; : > $out
run() { # label batch size sampler steps extra
  python bench.py --steps $5 --warmup 5 --batch $2 --size $3 --sampler $4 --no-secondary --no-cpu-baseline $6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 $6', round(d['ms_per_step'], 4), 'ms/step')" >> $out || return 1
}
for cfg in "c1 1 64 em 50" "b2 2 128 em 50" "b8 8 128 em 50" "c2 32 128 em 100" "b64 64 128 em 50" "c4 16 256 pc 30" "b4_256 4 256 em 30"; do
  set -- $cfg
  run $1 $2 $3 $4 $5 --no-autotune || exit 1
  if [ "$TUNED" = 1 ]; then run $1 $2 $3 $4 $5 || exit 1; fi
done
cat $out
