"""Which launches of a two-lane step run while BOTH lanes are idle (the serial sections: they count at face value)?
HIP events around every launch (ops profiler) + the stream each launch went to; the middle of three back-to-back steps."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops, profiler
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32


class P(profiler.Profiler):
    def end(self, tag, flops, nbytes, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((tag, flops, nbytes, start, e, torch.cuda.current_stream().cuda_stream))


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = torch.bfloat16; model.to(dev)
tr = SM3Trainer(model, lr=1e-6)
g = torch.Generator(device=dev).manual_seed(1)
derm = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
clinic = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
for _ in range(4): tr.step(derm, clinic)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record()
profs = []
for i in range(3):
    p = P(detail=False)
    ops.set_profiler(p); tr.step(derm, clinic); ops.set_profiler(None)
    profs.append(p)
torch.cuda.synchronize()
p = profs[1]
iv = [(base.elapsed_time(s), base.elapsed_time(e), tag, st) for tag, _, _, s, e, st in p.records]
t0 = min(a for a, b, t, s in iv); t1 = max(b for a, b, t, s in iv)
per_stream = collections.defaultdict(list)
for a, b, t, s in iv: per_stream[s].append((a - t0, b - t0, t))
order = sorted(per_stream, key=lambda s: -sum(b - a for a, b, _ in per_stream[s]))
print(f"step span {t1 - t0:.2f} ms; streams by busy time:")
for s in order:
    L = per_stream[s]
    print(f"   stream {s:#x}: {len(L):4d} launches, {sum(b - a for a, b, _ in L):7.2f} ms busy, first at {min(a for a, b, _ in L):6.2f}, last ends {max(b for a, b, _ in L):6.2f}")
lanes = order[:2]
others = order[2:]
# launches on the other streams
for s in others:
    c = collections.Counter(); n = collections.Counter()
    for a, b, t in per_stream[s]: c[t] += b - a; n[t] += 1
    print(f"stream {s:#x} (not a lane):")
    for t, v in c.most_common(12): print(f"   {t:24s} {n[t]:4d} launches {v:7.3f} ms")
# time when neither lane has a kernel in flight
ev = []
for s in lanes:
    for a, b, t in per_stream[s]: ev += [(a, 1), (b, -1)]
ev.sort(); depth = 0; last = 0.0; idle = 0.0; gaps = []
for t, d in ev:
    if depth == 0 and t - last > 0: idle += t - last; gaps.append((t - last, last, t))
    depth += d; last = t
print(f"neither lane has a kernel in flight for {idle:.2f} ms of {t1 - t0:.2f}; the ten longest such gaps:")
for dgap, a, b in sorted(gaps, reverse=True)[:10]:
    inside = [t for s in others for (x, y, t) in per_stream[s] if x < b and y > a]
    print(f"   {dgap*1e3:7.1f} us at {a:6.2f} ms: {collections.Counter(inside).most_common(4)}")
