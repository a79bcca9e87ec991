"""Two-lane timeline analysis of a rocprofv3 --kernel-trace CSV: busy union, overlap, gaps, per-kernel share."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# keep the last ~40% of the trace (steady state)
t_lo = rows[int(len(rows) * 0.55)][0]
rows = [r for r in rows if r[0] >= t_lo]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; last = t0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
span = t1 - t0
print(f"span {span/1e6:.1f} ms, >=1 kernel running {100*busy1/span:.1f} %, >=2 running {100*busy2/span:.1f} %, idle {100*(span-busy1)/span:.1f} %")
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    k = n.replace("void (anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0][:60]
    tot[k] += e - s; cnt[k] += 1
for k, v in tot.most_common(12):
    print(f"{v/1e6:9.2f} ms  {cnt[k]:6d}  avg {v/cnt[k]/1e3:8.1f} us  {k}")
