cd tools/micro/bin
for shape in "32 128 128 64 64" "32 64 64 64 64" "32 32 32 128 128" "32 16 16 256 256"; do
  echo "== $shape"
  ./w2x_bench $shape 2 2 3 0 10 | cut -c1-150
  ./w2x_bench $shape 9 2 3 0 10 | cut -c1-150
  ./w2x_bench $shape 2 2 3 0 10 | cut -c1-150
  ./w2x_bench $shape 9 2 3 0 10 | cut -c1-150
done
