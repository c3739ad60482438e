cd tools/micro/bin
for r in 1 2; do
for b in base st1 st2 st3 st4 st5; do
  echo "== $b"
  ./w2d_abl_$b 32 128 128 64 64 2 2 0 0 5 | cut -c1-120
  ./w2d_abl_$b 32 128 128 64 64 2 2 0 2 5 | cut -c1-120
done
done
