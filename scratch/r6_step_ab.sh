#!/bin/bash
# Step-level A/B of one environment switch on one box: usage  r6_step_ab.sh NAME VALUE_A VALUE_B [rounds]
# two-lane bench.py (20 steps) interleaved, then the single-lane kernel-time sums.
NAME=$1; A=$2; B=$3; ROUNDS=${4:-2}
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    env $NAME=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-lane $NAME=$v:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')"
  done
done
for v in $A $B; do
  env $NAME=$v python3 bench.py --single-lane --steps 4 --warmup 2 --no-cpu-baseline --breakdown /tmp/bd_$v.txt > /dev/null 2>&1
  echo "single-lane $NAME=$v:"; grep -E "conv_wgrad|sum of kernel" /tmp/bd_$v.txt
done
