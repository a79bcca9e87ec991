"""Reduce rocprofv3 --pmc passes (one directory per pass) of `bench.py --single-lane` to a per-kernel table:
MFMA-pipe busy share, wave occupancy-time, LDS bank-conflict share, instruction mix.  GRBM_GUI_ACTIVE is reported as the
sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS section), SQ_VALU_MFMA_BUSY_CYCLES as the sum over all 1 024 SIMDs.
Usage: collect_sq.py <out.txt> <pass_dir> [<pass_dir> ...]"""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # cycles the kernel ran (per XCD)
    if gui <= 0:
        continue
    n = calls[k]["GRBM_GUI_ACTIVE"]
    mfma = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)
    waves = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (gui * 256.0)      # quad-cycles -> average resident waves per CU
    conf = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    ldsb = 100.0 * c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui * 256.0)
    iv, il, im = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_LDS", 0.0), c.get("SQ_INSTS_MFMA", 0.0)
    rows.append((gui, k, n, mfma, waves, ldsb, conf, iv / max(im, 1.0), il / max(im, 1.0)))
rows.sort(reverse=True)
out = ["per-kernel SQ counters, one step of bench.py --single-lane (B=256, 224x224, bf16); separate --pmc passes",
       f"{'kernel':78s} {'calls':>5s} {'Mcycles':>8s} {'MFMA busy %':>11s} {'waves/CU':>8s} {'LDS busy %':>10s} {'LDS conflict %':>14s} {'VALU/MFMA':>9s} {'LDS/MFMA':>8s}"]
for gui, k, n, mfma, waves, ldsb, conf, vpm, lpm in rows[:28]:
    out.append(f"{k[:78]:78s} {n:5d} {gui / 1e6:8.2f} {mfma:11.1f} {waves:8.1f} {ldsb:10.1f} {conf:14.1f} {vpm:9.1f} {lpm:8.1f}")
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print("\n".join(out))
