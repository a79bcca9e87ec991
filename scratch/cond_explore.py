"""Where is a 16-bit step well conditioned?  From states of increasing training progress (f32 steps on one fixed batch of
learnable pairs, the T2 setting), one bf16 / fp16 step against the exact-f32 mode on the same state: gradient cosine,
|g| ratio, loss difference -- on the training batch and on a fresh one.  Chooses the state of the bf16 oracle anchor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_config_gpu as T
from sm3hip.trainer import SM3Trainer
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 224))
LR = float(os.environ.get("LR", 3e-4))
derm, clinic = T._latent_batch(B, S, 21)
fresh = T._latent_batch(B, S, 22)
model = T._build(21, torch.float32)
tr = SM3Trainer(model, lr=LR, weight_decay=5e-2, eps=1e-5, style=0)

def grad(dt, sd, d, c):
    m = T._build(0, dt, sd)
    t = SM3Trainer(m, lr=0.0, init_scale=1024.0)
    loss = float(t.step(d, c)); torch.cuda.synchronize()
    g = t._engine().store.flat_g.double().cpu()
    if dt == torch.float16:
        g = g / 1024.0 if t.steps_taken() == 1 else g * float("nan")
    del t, m
    return loss, g

for step in range(0, 13):
    if step in (0, 1, 2, 3, 4, 6, 8, 12):
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for name, (d, c) in (("train", (derm, clinic)), ("fresh", fresh)):
            l32, g32 = grad(torch.float32, sd, d, c)
            row = f"step {step:2d} {name}: loss f32 {l32:8.4f} |g| {float(g32.norm()):10.3f}"
            for dn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
                l, g = grad(dt, sd, d, c)
                cos = float(torch.dot(g, g32) / (g.norm() * g32.norm()))
                row += f" | {dn}: dloss {l - l32:+.4f} cos {cos:.3f} |g|/|g32| {float(g.norm() / g32.norm()):.3f}"
            print(row, flush=True)
    tr.step(derm, clinic)
