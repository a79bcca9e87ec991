"""Run-to-run spread of ONE step's gradient from identical state: per tensor, relative difference of two runs
(float atomics in the weight gradient reorder fp32 sums; anything beyond ~1e-5 of the tensor's norm would be a race)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
from test_config_gpu import _batch, _build
from sm3hip.trainer import SM3Trainer
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
B, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 64)
batch = _batch(B, S, 40)
runs = []
for r in range(3):
    m = _build(41, dt)
    tr = SM3Trainer(m, lr=0.0, **({"growth_interval": 3, "init_scale": 1024.0} if dt == torch.float16 else {}))
    loss = tr.step(*batch); torch.cuda.synchronize()
    st = tr._engine().store
    runs.append((float(loss), st.flat_g.clone(), st))
print("losses", [f"{l:.7f}" for l, _, _ in runs])
g0, g1, g2 = runs[0][1].double(), runs[1][1].double(), runs[2][1].double()
print(f"whole gradient: |g| {float(g0.norm()):.4e}  |g0-g1|/|g| {float((g0-g1).norm()/g0.norm()):.3e}  |g0-g2|/|g| {float((g0-g2).norm()/g0.norm()):.3e}")
st = runs[0][2]
rows = []
for name in st.offsets:
    a, b = st._view(g0, name).flatten(), st._view(g1, name).flatten()
    n = float(a.norm())
    rows.append((float((a - b).norm()) / (n + 1e-30), name, n, int(a.numel()), float((a - b).abs().max())))
rows.sort(reverse=True)
for rel, name, n, k, mx in rows[:25]: print(f"{rel:10.3e}  |g| {n:10.3e}  max abs diff {mx:9.3e}  n {k:8d}  {name}")
