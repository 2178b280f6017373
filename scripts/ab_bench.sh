#!/bin/bash
# A/B of two builds of the library on the SAME box: alternating bench runs (graph mode), ms_per_step of each.
# usage: scripts/ab_bench.sh scripts/_ab/libsaunet_base.so scripts/_ab/libsaunet_new.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "
    SAUNET_HIP_LIB=$PWD/$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
  done
done
