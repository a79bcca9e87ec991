#!/bin/bash
# usage: pmc_v2.sh <outdir>   (run on the GPU box from the repo root)
set -e
OUT=$1; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d $OUT/p$i -o pmc -- python3 scratch/pmc_v2.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm" not in k: continue
        name = "v2" if "v2" in k else "old"
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in agg:
    print(name)
    for c, v in sorted(agg[name].items()):
        print(f"  {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
