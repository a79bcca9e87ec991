#!/bin/bash
# SQ counters per kernel: four --pmc passes of the single-lane step (nothing else traced), reduced by collect_sq.py
set -e
OUT=gpurun_out/sq_$1; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $ctrs -d $OUT/p$i -o pmc -- python3 bench.py --single-lane --steps 1 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
python3 scratch/collect_sq.py $OUT/sq_per_kernel.txt $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
