"""Per-layer-group gradient cosine of one bf16 / fp16 step against the exact-f32 mode (B = 16, 224 x 224, learnable pairs):
where in the network does a 16-bit step keep its direction?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_config_gpu as T
from sm3hip.trainer import SM3Trainer
derm, clinic = T._latent_batch(16, 224, 21)
def grad(dt):
    m = T._build(21, dt)
    t = SM3Trainer(m, lr=0.0, init_scale=1024.0)
    t.step(derm, clinic); torch.cuda.synchronize()
    st = t._engine().store
    out = {n: st._view(st.flat_g, n).double().cpu().flatten() / (1024.0 if dt == torch.float16 else 1.0) for n in st.names}
    del t, m
    return out
g32 = grad(torch.float32)
groups = ["projector", "cross_proj", "layer4", "layer3", "layer2", "layer1", "encoder.conv1", "encoder.bn1"]
for dn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    g = grad(dt)
    print(dn)
    for grp in groups:
        for kind in ("conv", "bn", ""):
            names = [n for n in g if grp in n and ((kind == "conv" and "bn" not in n and "downsample.1" not in n and n.endswith("weight")) or
                                                     (kind == "bn" and ("bn" in n or "downsample.1" in n)) or (kind == "" and grp in ("projector", "cross_proj")))]
            if not names: continue
            a = torch.cat([g[n] for n in names]); b = torch.cat([g32[n] for n in names])
            print(f"   {grp:14s} {kind or 'all':5s} n={len(names):3d} cos {float(torch.dot(a, b) / (a.norm() * b.norm())):.3f} |g| {float(a.norm()):10.3f} / {float(b.norm()):10.3f}")
            if kind == "": break
