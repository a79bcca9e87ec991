#!/bin/bash
set -o pipefail
O=gpurun_out/r4e9; mkdir -p $O
for i in 1 2 3; do for w in 1 3; do
  SM3_CONV_W8=$w timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_w${w}_$i.json 2> $O/bench_w${w}_$i.err || { tail -5 $O/bench_w${w}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_w${w}_$i.json").read().strip().splitlines()[-1]); print("w8=$w run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"]["mfma_bound_launches"]["achieved_TFLOPs"], d["roofline"]["by_regime"]["hbm_bound_launches"]["achieved_GBs"])
PY
done; done
VARIANTS='[{"SM3_CONV_W8":"1"},{"SM3_CONV_W8":"3"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.2 > $O/ab_w8.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "K9x\|variants\|^tag\|sum of" $O/ab_w8.txt | head -40
timeout -k 10 900 python -m pytest tests/test_round3_gpu.py tests/test_dp_gpu.py -q -s -m gpu -k "T2_linear or 16bit_batchnorm" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|AUROC|^FAILED" $O/tests.log | tail
