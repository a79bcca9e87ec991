#!/bin/bash
set -o pipefail
O=gpurun_out/r4e5; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels.log 2>&1; echo "kernel+linbn tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/kernels.log
VARIANTS='[{"SM3_CONV_PW":"0"},{"SM3_CONV_PW":"1"},{"SM3_CONV_PW":"3"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.25 > $O/ab_pw.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/ab_pw.txt | head -75; tail -22 $O/ab_pw.txt
for i in 1 2; do for f in 0 3; do
  SM3_CONV_PW=$f timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_pw${f}_$i.json 2> $O/bench_pw${f}_$i.err || { tail -5 $O/bench_pw${f}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_pw${f}_$i.json").read().strip().splitlines()[-1]); print("pw=$f run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"])
PY
done; done
timeout -k 10 300 python scratch/pertensor_cos.py > $O/pertensor_cos.txt 2>&1; echo "pertensor rc=$?" | tee -a $O/summary.txt
cat $O/pertensor_cos.txt
