#!/bin/bash
set -o pipefail
O=gpurun_out/r4e11; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name FAILED"; tail -3 $O/bench_$name.err; return; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"])
PY
}
for i in 1 2; do
run base_$i A=1
run cap512_$i SM3_BN_GRID_CAP=512
run cap1024_$i SM3_BN_GRID_CAP=1024
run cap1536_$i SM3_BN_GRID_CAP=1536
run cap2048_$i SM3_BN_GRID_CAP=2048
run nopair_$i SM3_PAIR_VIEWS=0
run viewlanes_$i SM3_PAIR_VIEWS=0 SM3_VIEW_LANES=1
run unroll8_$i SM3_BN_UNROLL=8
run unroll2_$i SM3_BN_UNROLL=2
done
