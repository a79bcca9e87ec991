#!/bin/bash
set -o pipefail
O=gpurun_out/r4suite; mkdir -p $O
( while true; do sleep 60; echo "[alive] $(date +%T) $(tail -c 100 $O/gpu_tests.log | tr '\n' ' ')"; done ) &
KA=$!
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; rc=$?
kill $KA
echo "gpu tests rc=$rc"; tail -8 $O/gpu_tests.log
