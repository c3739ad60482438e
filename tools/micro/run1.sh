cd tools/micro/bin
for b in base; do
  echo "== $b"
  ./w2d_abl_$b 32 128 128 64 64 2 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 128 128 64 64 1 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 128 128 64 64 2 2 0 2 5 | cut -c1-140
  ./w2d_abl_$b 32 64 64 64 64 2 2 0 2 5 | cut -c1-140
  ./w2d_abl_$b 32 64 64 64 64 2 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 32 32 128 128 2 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 32 32 128 128 2 2 0 2 5 | cut -c1-140
  ./w2d_abl_$b 32 32 32 64 64 2 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 32 32 64 64 1 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 16 16 256 256 2 2 0 0 5 | cut -c1-140
  ./w2d_abl_$b 32 16 16 256 256 1 2 0 0 5 | cut -c1-140
done
