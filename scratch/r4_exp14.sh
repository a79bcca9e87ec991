#!/bin/bash
set -o pipefail
O=gpurun_out/r4e14; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/kernels.log
SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so timeout -k 10 300 python scratch/stamp_phases.py > $O/stamp_phases.txt 2>&1; echo "stamp phases rc=$?" | tee -a $O/summary.txt
grep -A8 "EPI 2" $O/stamp_phases.txt | head -24
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_$i.json 2> $O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1]); print("run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"]["mfma_bound_launches"]["achieved_TFLOPs"], d["roofline"]["by_regime"]["hbm_bound_launches"]["achieved_GBs"])
PY
done
timeout -k 10 200 python scratch/prof_detail.py 256 > $O/per_shape.txt 2>&1; head -40 $O/per_shape.txt
