#!/bin/bash
# same-box A/B of an environment switch (graph mode): scripts/ab_env.sh VAR [rounds]  -> ms_per_step with VAR=0 and VAR=1 alternating
V=$1; R=${2:-3}
for i in $(seq $R); do
  for x in 0 1; do
    echo -n "$V=$x "
    env $V=$x python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
  done
done
