#!/bin/bash
set -o pipefail
O=gpurun_out/r4halo; mkdir -p $O
VARIANTS='[{"SM3_CONV_DBG":"0"},{"SM3_CONV_DBG":"2"},{"SM3_CONV_DBG":"4"},{"SM3_CONV_DBG":"6"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.25 > $O/ab_staleA.txt 2>&1; echo rc=$?
grep "K9x" $O/ab_staleA.txt
