"""Per-shape A/B of whole training steps (lanes serialised, HIP events around every launch): kernel-level env switches
(read by the library at every launch) are flipped between steps of ONE process, R interleaved rounds.
  VARIANTS='[{"SM3_CONV_SPLIT":"0"},{"SM3_CONV_SPLIT":"1"}]' python scratch/ab_detail.py [B] [rounds] [min_ms]
Prints ms per step and tag for every variant (mean over rounds), sorted by the first variant's time."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops, profiler
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MINMS = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
variants = json.loads(os.environ.get("VARIANTS", "[{}]"))
keys = sorted({k for v in variants for k in v})
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = torch.bfloat16; model.to(dev)
tr = SM3Trainer(model, lr=1e-6)
tr._engine().two_streams = False
g = torch.Generator(device=dev).manual_seed(1)
derm = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
clinic = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
def setenv(v):
    for k in keys:
        if k in v: os.environ[k] = v[k]
        else: os.environ.pop(k, None)
for v in variants:
    setenv(v); tr.step(derm, clinic)
torch.cuda.synchronize()
acc = [dict() for _ in variants]
for r in range(R):
    for i, v in enumerate(variants):
        setenv(v)
        p = profiler.Profiler(detail=True)
        ops.set_profiler(p); tr.step(derm, clinic); torch.cuda.synchronize(); ops.set_profiler(None)
        for tag, d in p.summary().items():
            a = acc[i].setdefault(tag, {"ms": 0.0, "launches": d["launches"], "flops": d["flops"], "bytes": d["bytes"]})
            a["ms"] += d["ms"] / R
print("variants:", *[f"v{i}={json.dumps(v)}" for i, v in enumerate(variants)])
hdr = f"{'tag':52s} {'n':>4s} " + " ".join(f"{'v%d ms' % i:>8s}" for i in range(len(variants))) + "  " + " ".join(f"{'v%d/v0' % i:>6s}" for i in range(1, len(variants))) + f" {'TF(v0)':>7s} {'GB/s(v0)':>8s}"
print(hdr)
tags = sorted(acc[0], key=lambda t: -acc[0][t]["ms"])
cls = {}
for t in tags:
    ms = [a.get(t, {"ms": float('nan')})["ms"] for a in acc]
    c = cls.setdefault(t.split("|")[0], [0.0] * len(variants))
    for i, m in enumerate(ms): c[i] += m
    if ms[0] < MINMS: continue
    d = acc[0][t]
    print(f"{t[:52]:52s} {d['launches']:4d} " + " ".join(f"{m:8.3f}" for m in ms) + "  " + " ".join(f"{m / ms[0]:6.3f}" for m in ms[1:]) +
          f" {d['flops'] / ms[0] / 1e9:7.0f} {d['bytes'] / ms[0] / 1e6:8.0f}")
print("--- per class")
for c, ms in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    print(f"{c:52s}      " + " ".join(f"{m:8.3f}" for m in ms) + "  " + " ".join(f"{m / ms[0]:6.3f}" for m in ms[1:]))
tot = [sum(a[t]["ms"] for t in a) for a in acc]
print(f"{'sum of kernel time':52s}      " + " ".join(f"{m:8.3f}" for m in tot))
