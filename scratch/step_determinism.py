"""Two runs of the same fp16 steps from identical state: where do the PARAMETERS differ after step 1, 2?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
from test_config_gpu import _batch, _build
from sm3hip.trainer import SM3Trainer
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
batches = [_batch(8, 64, 40 + i) for i in range(2)]
snaps = []
for r in range(3):
    m = _build(41, dt)
    tr = SM3Trainer(m, lr=1e-4, **({"growth_interval": 3, "init_scale": 1024.0} if dt == torch.float16 else {}))
    out = []
    for i in range(2):
        loss = tr.step(*batches[i]); torch.cuda.synchronize()
        st = tr._engine().store
        out.append((float(loss), st.flat_p.clone().double(), st.flat_g.clone().double()))
    snaps.append((out, st))
st = snaps[0][1]
for i in range(2):
    print(f"after step {i + 1}: losses", [f"{s[0][i][0]:.6f}" for s in snaps])
    for a, b in ((0, 1), (0, 2)):
        pa, pb = snaps[a][0][i][1], snaps[b][0][i][1]
        d = (pa - pb).abs()
        print(f"  runs {a},{b}: params differing {int((d > 0).sum())} of {d.numel()}, > 1e-5: {int((d > 1e-5).sum())}, max {float(d.max()):.3e}")
    pa, pb = snaps[0][0][i][1], snaps[1][0][i][1]
    rows = []
    for name in st.offsets:
        x, y = st._view(pa, name).flatten(), st._view(pb, name).flatten()
        d = (x - y).abs()
        if float(d.max()) > 1e-5: rows.append((int((d > 1e-5).sum()), float(d.max()), name, x.numel()))
    rows.sort(reverse=True)
    for n, mx, name, k in rows[:12]: print(f"     {n:8d} of {k:8d} elements differ by > 1e-5 (max {mx:.2e})  {name}")
