#!/bin/bash
# round-4 experiment 1: loader/consumer split -- kernel tests, per-shape A/B, whole-step A/B
set -o pipefail
O=gpurun_out/r4e1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/kernel_tests.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/kernel_tests.log | tee -a $O/summary.txt
grep -q "passed" $O/kernel_tests.log || exit 1
FWD_VARIANTS='[{"SM3_CONV_SPLIT":"0"},{"SM3_CONV_SPLIT":"1","SM3_CONV_DBG":"0"},{"SM3_CONV_SPLIT":"1","SM3_CONV_DBG":"1"}]' timeout -k 10 300 python scratch/bench_kernels.py fwd > $O/fwd_ab.txt 2>&1; echo "fwd ab rc=$?" | tee -a $O/summary.txt
cat $O/fwd_ab.txt
for i in 1 2; do
for sp in 0 1; do
  SM3_CONV_SPLIT=$sp timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-dtypes > $O/bench_split${sp}_$i.json 2> $O/bench_split${sp}_$i.err || { echo "bench split=$sp failed"; tail -5 $O/bench_split${sp}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_split${sp}_$i.json").read().strip().splitlines()[-1]); print("split=$sp run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
