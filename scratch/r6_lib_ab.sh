#!/bin/bash
# Step-level A/B of two builds of the library on one box: usage  r6_lib_ab.sh LIB_A LIB_B [rounds]   (SM3_LIBRARY selects the .so)
A=$1; B=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    SM3_LIBRARY=$PWD/$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-lane $v:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')"
  done
done
for v in $A $B; do
  SM3_LIBRARY=$PWD/$v python3 bench.py --single-lane --steps 4 --warmup 2 --no-cpu-baseline --breakdown /tmp/bd.txt > /dev/null 2>&1
  echo "single-lane $v:"; grep -E "conv_gemm|conv_wgrad|sum of kernel" /tmp/bd.txt
done
