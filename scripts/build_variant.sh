#!/bin/bash
# A/B helper: rebuild ONE translation unit with extra flags and link it with the other (current) objects into a variant library.
# usage: scripts/build_variant.sh scripts/_ab/libX.so conv_tile.hip -DSAUNET_WGRAD_PREFETCH=0     (run python -m saunet_amd._build first)
set -e
OUT=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); P=$R/shape-attentive-unet_amd
mkdir -p /tmp/_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $P/csrc/$SRC -o /tmp/_variant/${SRC%.hip}.o
OBJS=""
for o in $P/_obj/*.o; do b=$(basename $o); if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS /tmp/_variant/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$OUT $OBJS
echo $OUT
