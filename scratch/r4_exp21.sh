#!/bin/bash
set -o pipefail
O=gpurun_out/r4e21; mkdir -p $O
SM3_WGRAD_TAP1=2 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "wgrad or conv_fwd" > $O/wgrad_tests.log 2>&1; echo "wgrad tests (KG=2 tap) rc=$?" | tee -a $O/summary.txt; tail -2 $O/wgrad_tests.log
VARIANTS='[{},{"SM3_WGRAD_TAP1":"2"},{"SM3_WGRAD_TAP1":"1"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.1 > $O/ab_wgrad2.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_wgrad|.*K9\|conv_wgrad|.*K[24]x\|variants\|^tag\|sum of\|^conv_wgrad " $O/ab_wgrad2.txt | head -20
