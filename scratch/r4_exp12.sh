#!/bin/bash
set -o pipefail
O=gpurun_out/r4e12; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name FAILED"; tail -3 $O/bench_$name.err; return; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"])
PY
}
for i in 1 2; do
run base_$i A=1
run cap512_$i SM3_BN_GRID_CAP=512
run cap384_$i SM3_BN_GRID_CAP=384
run cap256_$i SM3_BN_GRID_CAP=256
run cap192_$i SM3_BN_GRID_CAP=192
run cap512u2_$i SM3_BN_GRID_CAP=512 SM3_BN_UNROLL=2
run cap384u8_$i SM3_BN_GRID_CAP=384 SM3_BN_UNROLL=8
run cap256u8_$i SM3_BN_GRID_CAP=256 SM3_BN_UNROLL=8
done
