#!/bin/bash
set -o pipefail
O=gpurun_out/r4e17; mkdir -p $O
SM3_LIBRARY=scratch/_occ/libsm3hip_occ.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "wgrad" > $O/wgrad_tests.log 2>&1; echo "wgrad tests (occ lib) rc=$?" | tee -a $O/summary.txt; tail -2 $O/wgrad_tests.log
for lib in base occ; do
  L=""; [ $lib = occ ] && L=scratch/_occ/libsm3hip_occ.so
  SM3_LIBRARY=$L timeout -k 10 200 python scratch/prof_detail.py 256 > $O/per_shape_$lib.txt 2>&1
  echo "== $lib"; grep "conv_wgrad|.*K9x\|conv_wgrad|.*K[24]x\|sum of" $O/per_shape_$lib.txt
done
for i in 1 2 3; do for lib in base occ; do
  L=""; [ $lib = occ ] && L=scratch/_occ/libsm3hip_occ.so
  SM3_LIBRARY=$L timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-dtypes > $O/bench_${lib}_$i.json 2> $O/bench_${lib}_$i.err || { tail -5 $O/bench_${lib}_$i.err; exit 1; }
  python - <<PY | tee -a $O/summary.txt
import json; d=json.loads(open("$O/bench_${lib}_$i.json").read().strip().splitlines()[-1]); print("lib=$lib run $i", d["value"], d["ms_per_step"])
PY
done; done
