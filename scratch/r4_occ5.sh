#!/bin/bash
# A/B of the 96-register plain-forward build (scratch/_occ5) against the shipped library: co-run pairs + the 20-step bench.
set -o pipefail
O=gpurun_out/r4occ5; mkdir -p $O
timeout -k 10 200 python scratch/corun.py > $O/corun_base.txt 2>&1 && \
SM3_LIBRARY=$PWD/scratch/_occ5/libsm3hip_occ5.so timeout -k 10 200 python scratch/corun.py > $O/corun_occ5.txt 2>&1 && \
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_base_$i.json && \
  SM3_LIBRARY=$PWD/scratch/_occ5/libsm3hip_occ5.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_occ5_$i.json || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4occ5/bench_*.json")):
    r = json.loads(open(f).read()); print(f, r["value"], r["ms_per_step"])
PY
