#!/bin/bash
set -o pipefail
O=gpurun_out/r4e13; mkdir -p $O
SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so timeout -k 10 300 python scratch/stamp_conv.py > $O/stamp_kloop_1stage.txt 2>&1; echo "stamp 1-stage rc=$?" | tee -a $O/summary.txt
SM3_CONV_SINGLE_STAGE_MAX=8 SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so timeout -k 10 300 python scratch/stamp_conv.py > $O/stamp_kloop_2stage.txt 2>&1; echo "stamp 2-stage rc=$?" | tee -a $O/summary.txt
SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so timeout -k 10 300 python scratch/stamp_phases.py > $O/stamp_phases.txt 2>&1; echo "stamp phases rc=$?" | tee -a $O/summary.txt
cat $O/stamp_kloop_1stage.txt; grep -A12 "layer3 conv2" $O/stamp_kloop_2stage.txt | head -16
