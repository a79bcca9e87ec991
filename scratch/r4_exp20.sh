#!/bin/bash
set -o pipefail
O=gpurun_out/r4e20; mkdir -p $O
SM3_WGRAD_TAP1=1 SM3_WGRAD_KG=1 SM3_WGRAD_DENSE_NST=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu -k "wgrad or slab or cat or linbn or conv_fwd" > $O/wgrad_tests.log 2>&1; echo "wgrad tests (1-stage variants) rc=$?" | tee -a $O/summary.txt; tail -2 $O/wgrad_tests.log
VARIANTS='[{},{"SM3_WGRAD_TAP1":"1"},{"SM3_WGRAD_KG":"1","SM3_WGRAD_DENSE_NST":"1"},{"SM3_WGRAD_TAP1":"1","SM3_WGRAD_KG":"1","SM3_WGRAD_DENSE_NST":"1"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.1 > $O/ab_wgrad1.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_wgrad\|variants\|^tag\|sum of" $O/ab_wgrad1.txt | head -50
