#!/bin/bash
# Debug aid: builds variants of the align kernel (extra -D flags) next to the product library
# and times S4 with each (run the timing part through gpurun).  Usage:
#   scripts/align_variants.sh build "name1:-DFLAG1" "name2:-DFLAG2 -DFLAG3" ...
#   scripts/align_variants.sh time name1 name2 ...
set -eu
cd "$(dirname "$0")/../whisper_amd/csrc"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p exp
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden $flags -c myers_align.hip -o exp/align_$name.o
    objs=$(ls *.o | grep -v myers_align.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libwamd_$name.so $objs exp/align_$name.o
  done
else
  cd ../..
  for name in "$@"; do
    WAMD_LIB=whisper_amd/csrc/exp/libwamd_$name.so timeout -k 10 120 python3 scripts/time_align.py $name
  done
fi
