#!/bin/bash
set -o pipefail
O=gpurun_out/r4e16; mkdir -p $O
for i in 1 2 3; do for m in all last none; do
  SM3_BENCH_REGION_EVENTS=$m timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_${m}_$i.json 2> $O/bench_${m}_$i.err || { tail -5 $O/bench_${m}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_${m}_$i.json").read().strip().splitlines()[-1]); print("region_events=$m run $i", d["value"], d["ms_per_step"], d["roofline"]["frac_in_timed_region_two_lanes"])
PY
done; done
