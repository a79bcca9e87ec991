"""Per-tensor comparison of the bf16 and exact-f32 gradients of one step (same weights, same batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_config_gpu as T
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(5)
model = SimCLRSkinV32("resnet50", None, 128, 0.1)
model.sm3_dtype = torch.float32
model.to(T.DEV)
tr = SM3Trainer(model, lr=3e-4, weight_decay=5e-2, eps=1e-5, style=0)
derm, clinic = T._latent_batch(B, 64, 7)
for _ in range(steps):
    tr.step(derm, clinic)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
res = {}
for tag, dt in (("f32", torch.float32), ("f32b", torch.float32), ("bf16", torch.bfloat16)):
    m = T._build(0, dt, sd); m.train()
    res[tag] = T._compat_step(m, derm, clinic)[2]
    del m
rows = []
for k in res["f32"]:
    a, b, c = res["f32"][k].double().flatten(), res["bf16"][k].double().flatten(), res["f32b"][k].double().flatten()
    rows.append((float(a.norm()), float(b.norm()), float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a @ c / (a.norm() * c.norm() + 1e-30)), k))
rows.sort(reverse=True)
print("top tensors by f32 gradient norm:  |g32|  |gbf16|  cos(f32,bf16)  cos(f32,f32 rerun)")
for r in rows[:25]:
    print(f"{r[0]:12.4f} {r[1]:12.4f} {r[2]:8.4f} {r[3]:8.4f}  {r[4]}")
import statistics
print("median cosine over tensors:", statistics.median(r[2] for r in rows))
for pat in ("conv", "bn", "downsample.1", "projector", "cross_proj"):
    sel = [r for r in rows if pat in r[4]]
    print(pat, "n", len(sel), "median cos", statistics.median(r[2] for r in sel), "min", min(r[2] for r in sel))
A = torch.cat([v.double().flatten() for v in res["f32"].values()]); Bv = torch.cat([v.double().flatten() for v in res["bf16"].values()])
print("global cosine", float(A @ Bv / (A.norm() * Bv.norm())))
