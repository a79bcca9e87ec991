#!/bin/bash
set -o pipefail
O=gpurun_out/r4halo; mkdir -p $O
for i in 1 2; do for f in 0 1; do
  SM3_CONV_HALO=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes > $O/bench_halo${f}_$i.json 2> $O/bench_halo${f}_$i.err || { tail -5 $O/bench_halo${f}_$i.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_halo${f}_$i.json").read().strip().splitlines()[-1]); print("halo=$f run $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("by_regime"))
PY
done; done
SM3_CONV_HALO=1 timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py -q -x -m gpu > $O/tests_halo1.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_halo1.log
