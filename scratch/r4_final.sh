#!/bin/bash
# final validation of the round: smoke, whole GPU suite, evidence set r04c
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4final; mkdir -p $O
( while true; do sleep 60; echo "[alive] $(date +%T)"; done ) &
KA=$!
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -1 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/gpu_tests.log
bash scratch/collect_profiles.sh r04c > $O/collect.log 2>&1; echo "collect rc=$?" | tee -a $O/summary.txt
kill $KA
tail -c 700 gpurun_out/prof_r04c/bench.json; echo
