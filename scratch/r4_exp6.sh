#!/bin/bash
set -o pipefail
O=gpurun_out/r4e6; mkdir -p $O
VARIANTS='[{"SM3_CONV_SINGLE_STAGE_MAX":"8"},{"SM3_CONV_SINGLE_STAGE_MAX":"5"},{"SM3_CONV_SINGLE_STAGE_MAX":"3"},{"SM3_CONV_SINGLE_STAGE_MAX":"1"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.2 > $O/ab_stage.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_gemm\|variants\|^tag\|sum of" $O/ab_stage.txt
DC=1.0 timeout -k 10 300 python scratch/t2_stream3.py 2>&1 | grep -v Warn | tee $O/t2_dc1.txt
DC=2.0 timeout -k 10 300 python scratch/t2_stream3.py 2>&1 | grep -v Warn | tee $O/t2_dc2.txt
