#!/bin/bash
set -o pipefail
O=gpurun_out/r4e8; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/kernels.log
SM3_CONV_SINGLE_STAGE_MAX=1000 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_linbn_gpu.py -q -x -m gpu > $O/kernels_1stage.log 2>&1; echo "kernel tests (1-stage everywhere) rc=$?" | tee -a $O/summary.txt; tail -2 $O/kernels_1stage.log
for i in 1 2; do for m in 8 20 40 80 1000; do
  SM3_CONV_SINGLE_STAGE_MAX=$m timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_m${m}_$i.json 2> $O/bench_m${m}_$i.err || { tail -5 $O/bench_m${m}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_m${m}_$i.json").read().strip().splitlines()[-1]); print("single_stage_max=$m run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_regime"]["mfma_bound_launches"]["achieved_TFLOPs"], d["roofline"]["by_regime"]["hbm_bound_launches"]["achieved_GBs"])
PY
done; done
for i in 1 2; do for sw in 0 1; do
  GPU_MAX_HW_QUEUES=8 SM3_SIDE_WGRAD=$sw SM3_CONV_SINGLE_STAGE_MAX=40 timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_sw${sw}_$i.json 2> $O/bench_sw${sw}_$i.err || { tail -5 $O/bench_sw${sw}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_sw${sw}_$i.json").read().strip().splitlines()[-1]); print("hwq8 side_wgrad=$sw run $i", d["value"], d["ms_per_step"])
PY
done; done
VARIANTS='[{"SM3_CONV_SINGLE_STAGE_MAX":"40"},{"SM3_CONV_SINGLE_STAGE_MAX":"80"},{"SM3_CONV_SINGLE_STAGE_MAX":"1000"}]' timeout -k 10 400 python scratch/ab_detail.py 256 3 0.2 > $O/ab_stage_up2.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep "conv_gemm\|variants\|^tag\|sum of" $O/ab_stage_up2.txt | head -40
