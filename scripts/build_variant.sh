#!/bin/bash
# Builds path-tracer_amd/variants/<name>.so: libptk with ONE translation unit recompiled with
# extra flags (measurement builds; scripts/gpu_dn_variants.py / gpu_variants.py time them).
# Usage: scripts/build_variant.sh <name> <unit, e.g. denoise.hip> "<extra flags>"
set -e
cd "$(dirname "$0")/../path-tracer_amd"
NAME=$1; UNIT=$2; EXTRA=$3
make -s -j8 libptk.so
mkdir -p variants build/variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable -I../include"
XHIP=""; case "$UNIT" in *.cpp) XHIP="-x hip";; esac
/opt/rocm/bin/hipcc $FLAGS $EXTRA $XHIP -c csrc/$UNIT -o build/variants/$NAME.o
OBJS=""
for o in build/*.o; do
  if [ "$(basename $o)" = "$UNIT.o" ]; then OBJS="$OBJS build/variants/$NAME.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$NAME.so $OBJS
echo "built variants/$NAME.so"
