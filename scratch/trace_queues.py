"""Per-hardware-queue summary of a rocprofv3 kernel trace: kernels, busy time, and how much of the time two queues ran at once."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = []
per = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r[key]
    d = per.setdefault(q, [0, 0, {}])
    d[0] += 1; d[1] += e - s
    nm = r["Kernel_Name"][:40]
    d[2][nm] = d[2].get(nm, 0) + 1
    ev.append((s, 1)); ev.append((e, -1))
for q, (n, busy, names) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print(f"queue {q}: {n} kernels, busy {busy/1e6:.1f} ms, e.g. {top}")
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = sum(hist.values())
print("time with k kernels in flight:", {k: f"{100*v/tot:.1f}%" for k, v in sorted(hist.items())})
