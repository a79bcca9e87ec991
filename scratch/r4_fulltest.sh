#!/bin/bash
set -o pipefail
O=gpurun_out/r4full; mkdir -p $O
timeout -k 10 300 python scratch/halo_check.py 2>&1 | grep -v amdgpu.ids > $O/halo_check.txt; echo "halo check rc=$?"; tail -5 $O/halo_check.txt
( while true; do sleep 60; echo "[alive] $(date +%T) $(tail -c 120 $O/gpu_tests.log | tr '\n' ' ')"; done ) &
KA=$!
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; rc=$?
kill $KA
echo "gpu tests rc=$rc" | tee -a $O/summary.txt
tail -5 $O/gpu_tests.log
