#!/bin/bash
# upper bound of "fewer L2 -> LDS bytes" for the tap-shifted weight gradient: stale dY / X / both on 2 of 3 K-steps
set -o pipefail
O=gpurun_out/r4wst; mkdir -p $O
timeout -k 10 200 python3 scratch/prof_detail.py 256 2>/dev/null | grep "conv_wgrad|M.*K9x" > $O/base.txt || exit 1
for v in 1 2 3; do
  SM3_LIBRARY=$PWD/scratch/_wst/libsm3hip_wst$v.so timeout -k 10 200 python3 scratch/prof_detail.py 256 2>/dev/null | grep "conv_wgrad|M.*K9x" > $O/v$v.txt || exit 1
done
paste $O/base.txt $O/v1.txt $O/v2.txt $O/v3.txt | awk '{printf "%-40s n=%s base %s ms | stale dY %s | stale X %s | both %s\n", $1, $2, $3, $9, $15, $21}'
