set -e
cp profiles/r03_c2_tiles.txt /tmp/t.txt
for m in 4096 2048 512; do
  echo "== min_m $m"
  SBGM_ATTN_FUSED_MIN_M=$m python3 bench.py --steps 100 --warmup 10 --tune-cache /tmp/t.txt --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200
done
