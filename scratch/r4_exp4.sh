#!/bin/bash
set -o pipefail
O=gpurun_out/r4e4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_round4_gpu.py -q -s -m gpu -k "bn_stats_finalize" > $O/fused_stats_test.log 2>&1; echo "fused stats test rc=$?" | tee -a $O/summary.txt
tail -3 $O/fused_stats_test.log
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -q -x -m gpu > $O/e2e.log 2>&1; echo "e2e+kernels rc=$?" | tee -a $O/summary.txt
tail -3 $O/e2e.log
for i in 1 2; do for f in 0 1; do
  SM3_BN_FUSED_STATS=$f timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_fused${f}_$i.json 2> $O/bench_fused${f}_$i.err || { tail -5 $O/bench_fused${f}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_fused${f}_$i.json").read().strip().splitlines()[-1]); print("fused_stats=$f run $i", d["value"], d["ms_per_step"])
PY
done; done
timeout -k 10 300 python scratch/cond_explore.py > $O/cond_explore.txt 2>&1; echo "cond rc=$?" | tee -a $O/summary.txt
cat $O/cond_explore.txt
timeout -k 10 500 python scratch/t2_stream2.py > $O/t2_stream2.txt 2>&1; echo "t2 rc=$?" | tee -a $O/summary.txt
cat $O/t2_stream2.txt
