#!/bin/bash
set -o pipefail
O=gpurun_out/r4wprep; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_round4_gpu.py -m gpu -q -x -k "weight_prep" 2>&1 | tail -4
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes 2>/dev/null | tail -1 > $O/b_$i.json && python - <<PY
import json; d=json.loads(open("$O/b_$i.json").read()); print("bench", d["value"], d["ms_per_step"])
PY
done
python bench.py --single-lane --steps 4 --warmup 2 --no-cpu-baseline --no-other-dtypes --breakdown $O/bd.txt > /dev/null 2>&1; grep "weight_prep\|sum of" $O/bd.txt
