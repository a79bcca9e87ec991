#!/bin/bash
set -o pipefail
O=gpurun_out/r4e7; mkdir -p $O
VARIANTS='[{"SM3_CONV_SINGLE_STAGE_MAX":"8"},{"SM3_CONV_SINGLE_STAGE_MAX":"12"},{"SM3_CONV_SINGLE_STAGE_MAX":"20"},{"SM3_CONV_SINGLE_STAGE_MAX":"40"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.2 > $O/ab_stage_up.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_gemm\|variants\|^tag\|sum of" $O/ab_stage_up.txt
for i in 1 2 3; do for lib in base pf; do
  L=""; [ $lib = pf ] && L=scratch/_pf/libsm3hip_pf.so
  SM3_LIBRARY=$L timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_${lib}_$i.json 2> $O/bench_${lib}_$i.err || { tail -5 $O/bench_${lib}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_${lib}_$i.json").read().strip().splitlines()[-1]); print("lib=$lib run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"]["mfma_bound_launches"])
PY
done; done
timeout -k 10 900 python -m pytest tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_dp_gpu.py -q -s -m gpu -k "T2 or 16bit or forms or 16bit_batchnorm or fp16_mode" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|AUROC|224x224|448x448|B=16|rank [01]:|^FAILED" $O/tests.log | tail -30
