#!/bin/bash
# Debug aid: builds variants of the distance kernels (extra -D flags) next to the product
# library and times them (run the timing part through gpurun).  Usage:
#   scripts/dist_variants.sh build "name1:-DFLAG1" "name2:-DFLAG2 -DFLAG3" ...
#   scripts/dist_variants.sh time name1 name2 ...
set -eu
cd "$(dirname "$0")/../whisper_amd/csrc"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p exp
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden $flags -c myers_distance.hip -o exp/dist_$name.o
    objs=$(ls *.o | grep -v myers_distance.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libwamd_$name.so $objs exp/dist_$name.o
  done
else
  cd ../..
  for name in "$@"; do
    WAMD_LIB=whisper_amd/csrc/exp/libwamd_$name.so timeout -k 10 120 python3 scripts/time_band.py $name
  done
fi
