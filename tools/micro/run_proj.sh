cd tools/micro/bin
export W2D_PROJ=1
for r in 1 2; do
for b in old new; do
  echo "== $b"
  ./w2d_$b 32 128 128 64 64 2 2 3 2 10 | cut -c1-150
  ./w2d_$b 16 256 256 64 64 2 2 3 2 5 | cut -c1-150
done
done
