#!/bin/bash
set -o pipefail
O=gpurun_out/r4e2; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_p2p_gpu.py tests/test_abi.py -x -q -s > $O/p2p_tests.log 2>&1; echo "p2p tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/p2p_tests.log | tee -a $O/summary.txt
VARIANTS='[{"SM3_CONV_SPLIT":"0"},{"SM3_CONV_SPLIT":"1"},{"SM3_CONV_SPLIT":"3"},{"SM3_CONV_SPLIT":"5"},{"SM3_CONV_SPLIT":"9"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.3 > $O/ab_split.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/ab_split.txt
