#!/bin/bash
# Builds path-tracer_amd/variants/<name>.so: libptk with SEVERAL translation units recompiled with
# extra flags (for build flags that the kernels and the host side both read).
# Usage: scripts/build_variant_units.sh <name> "<extra flags>" <unit> [<unit> ...]
set -e
cd "$(dirname "$0")/../path-tracer_amd"
NAME=$1; EXTRA=$2; shift 2
make -s -j8 libptk.so
mkdir -p variants build/variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable -I../include"
for UNIT in "$@"; do
  XHIP=""; case "$UNIT" in *.cpp) XHIP="-x hip";; esac
  /opt/rocm/bin/hipcc $FLAGS $EXTRA $XHIP -c csrc/$UNIT -o build/variants/$NAME.$UNIT.o &
done
wait
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  if [ -f build/variants/$NAME.$b.o ]; then OBJS="$OBJS build/variants/$NAME.$b.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$NAME.so $OBJS
echo "built variants/$NAME.so"
