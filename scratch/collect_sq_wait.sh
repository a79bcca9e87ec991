#!/bin/bash
# Where the cycles of the production GEMM kernels go: SQ wait / issue-stall buckets per kernel.
# Separate --pmc passes of the single-lane step, nothing else traced; reduced by collect_sq_wait.py.
# usage: collect_sq_wait.sh <tag>   (repo root on the GPU box)
OUT=gpurun_out/sqw_$1; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1   # what bench.py sets itself; under rocprofv3 the runtime is up before the script runs
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  if rocprofv3 --output-format csv --pmc $ctrs -d $OUT/p$i -o pmc -- python3 bench.py --single-lane --steps 1 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1; then
    echo "pass $i done"
  else
    echo "pass $i FAILED ($ctrs)"; tail -5 $OUT/p$i.log
  fi
done
python3 scratch/collect_sq_wait.py $OUT/sq_wait_per_kernel.txt $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
