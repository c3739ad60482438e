#!/bin/bash
# builds the timing variants of tools/micro/w2d_exp.hip (development only)
set -e
cd "$(dirname "$0")/../.."
H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -I sbgm_danra_amd/csrc"
mkdir -p tools/micro/bin
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  [ "$defs" = "$v" ] && defs=""
  $H -DKFILE='"w2d_exp.hip"' $defs tools/micro/w2d_bench.hip -o tools/micro/bin/w2d_$name &
done
wait
ls tools/micro/bin
