#!/bin/bash
set -o pipefail
O=gpurun_out/r4e18; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/kernels.log
grep -q " passed" $O/kernels.log || { tail -30 $O/kernels.log; exit 1; }
VARIANTS='[{"SM3_CONV_PW":"3"},{"SM3_CONV_PW":"7"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.3 > $O/ab_noadd.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "_fz \|variants\|^tag\|sum of\|^conv_gemm" $O/ab_noadd.txt | head -40
for i in 1 2 3; do for m in 3 7; do
  SM3_CONV_PW=$m timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_pw${m}_$i.json 2> $O/bench_pw${m}_$i.err || { tail -5 $O/bench_pw${m}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_pw${m}_$i.json").read().strip().splitlines()[-1]); print("pw=$m run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
