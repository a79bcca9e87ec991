#!/bin/bash
set -o pipefail
O=gpurun_out/r4e10; mkdir -p $O
VARIANTS='[{},{"SM3_WGRAD_KG":"1"},{"SM3_WGRAD_KG":"1","SM3_WGRAD_DENSE_NST":"2"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.1 > $O/ab_wgrad.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_wgrad\|variants\|^tag\|sum of" $O/ab_wgrad.txt | head -60
