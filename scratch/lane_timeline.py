"""Two-lane timeline of ONE real step from the HIP events the ops wrappers record around every launch (the lanes run as in the
timed region: two streams, nothing serialised): how much of the step has 0 / 1 / >= 2 kernels in flight, and which kernel
classes run alone.  (rocprofv3 --kernel-trace cannot show this: it serialises dispatches.)"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops, profiler
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
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
# three instrumented steps enqueued back to back (the host runs ahead, as in the timed region); the MIDDLE one is analysed
base = torch.cuda.Event(enable_timing=True); base.record()
profs = []
for i in range(3):
    p = profiler.Profiler(detail=False)
    ops.set_profiler(p); tr.step(derm, clinic); ops.set_profiler(None)
    profs.append(p)
torch.cuda.synchronize()
p = profs[1]
iv = [(base.elapsed_time(s), base.elapsed_time(e), tag) for tag, _, _, s, e in p.records]
t0 = min(a for a, b, t in iv); t1 = max(b for a, b, t in iv)
iv = [(a - t0, b - t0, t) for a, b, t in iv]
span = t1 - t0
ev = sorted([(a, 1, t) for a, b, t in iv] + [(b, -1, t) for a, b, t in iv])
busy = [0.0, 0.0, 0.0]; alone = collections.Counter(); depth = 0; last = 0.0; running = collections.Counter()
for t, d, tag in ev:
    dt = t - last
    busy[min(depth, 2)] += dt
    if depth == 1:
        alone[next(k for k, v in running.items() if v > 0)] += dt
    depth += d; running[tag] += d; last = t
busy[0] += span - last
print(f"one instrumented two-lane step: {span:.2f} ms ({len(iv)} launches; the events add ~1 ms); kernels in flight: "
      f"0 for {busy[0]:.2f} ms ({100*busy[0]/span:.1f} %), 1 for {busy[1]:.2f} ms ({100*busy[1]/span:.1f} %), >= 2 for {busy[2]:.2f} ms ({100*busy[2]/span:.1f} %)")
print(f"sum of kernel durations under contention {sum(b - a for a, b, _ in iv):.1f} ms")
print("time with exactly ONE kernel in flight, by the class that runs alone:")
for k, v in alone.most_common(10):
    print(f"   {k:24s} {v:6.2f} ms")
