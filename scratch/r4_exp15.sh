#!/bin/bash
set -o pipefail
O=gpurun_out/r4e15; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels.log 2>&1; echo "kernel tests (M16 on) rc=$?" | tee -a $O/summary.txt; tail -2 $O/kernels.log
grep -q passed $O/kernels.log || { tail -40 $O/kernels.log; exit 1; }
VARIANTS='[{"SM3_CONV_M16":"0"},{"SM3_CONV_M16":"1"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.3 > $O/ab_m16.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_gemm\|variants\|^tag\|sum of" $O/ab_m16.txt | head -50
for i in 1 2 3; do for m in 0 1; do
  SM3_CONV_M16=$m timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_m${m}_$i.json 2> $O/bench_m${m}_$i.err || { tail -5 $O/bench_m${m}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_m${m}_$i.json").read().strip().splitlines()[-1]); print("m16=$m run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"]["mfma_bound_launches"]["achieved_TFLOPs"], d["roofline"]["by_regime"]["hbm_bound_launches"]["achieved_GBs"])
PY
done; done
