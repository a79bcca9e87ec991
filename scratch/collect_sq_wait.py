"""Reduce the --pmc passes of collect_sq_wait.sh to a per-kernel table of where the wave cycles go.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over all waves; WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots).  SQ_VALU_MFMA_BUSY_CYCLES is in cycles summed
over the 1 024 SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs.
Usage: collect_sq_wait.py <out.txt> <pass_dir> [...]"""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
rows = []
for k, c in acc.items():
    n = max(cnt[k].values())
    passes = max(1, round(cnt[k]["GRBM_GUI_ACTIVE"] / max(1, min(v for v in cnt[k].values()))))
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / passes
    if gui <= 0:
        continue
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    pct = lambda name: 100.0 * c.get(name, 0.0) / wc
    mfma = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)
    coex = 100.0 * c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / (gui * 1024.0)
    waves = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (gui * 256.0)
    im = max(c.get("SQ_INSTS_MFMA", 0.0), 1.0)
    rows.append((gui, k, min(cnt[k].values()), mfma, coex, waves, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_WAIT_INST_LDS"),
                 pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"), pct("SQ_ACTIVE_INST_VMEM"),
                 c.get("SQ_INSTS_VALU", 0) / im, c.get("SQ_INSTS_LDS", 0) / im, c.get("SQ_INSTS_VMEM", 0) / im,
                 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)))
rows.sort(reverse=True)
hdr = (f"{'kernel':74s} {'calls':>5s} {'Mcyc':>7s} {'MFMAbusy%':>9s} {'coexec%':>7s} {'waves/CU':>8s} | share of SQ_WAVE_CYCLES: "
       f"{'WAIT_ANY':>8s} {'WAIT_INST':>9s} {'(INST_LDS)':>10s} {'ACTIVE':>6s} {'aVALU':>6s} {'aLDS':>5s} {'aVMEM':>6s} | "
       f"{'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'VMEM/MFMA':>9s} {'LDSconf%':>8s}")
out = ["per-kernel SQ wait buckets, one step of bench.py --single-lane (B=256, 224x224, bf16); separate --pmc passes", hdr]
for r in rows[:30]:
    gui, k, n = r[0], r[1], r[2]
    out.append(f"{k[:74]:74s} {n:5d} {gui / 1e6:7.2f} {r[3]:9.1f} {r[4]:7.1f} {r[5]:8.1f} |" + " " * 27 +
               f"{r[6]:8.1f} {r[7]:9.1f} {r[8]:10.1f} {r[9]:6.1f} {r[10]:6.1f} {r[11]:5.1f} {r[12]:6.1f} | "
               f"{r[13]:9.1f} {r[14]:8.1f} {r[15]:9.1f} {r[16]:8.1f}")
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print("\n".join(out))
