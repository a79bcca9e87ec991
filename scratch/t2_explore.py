"""bf16 vs exact-f32 step from states of increasing training progress (choosing T2's state and bounds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_config_gpu as T
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(5)
model = SimCLRSkinV32("resnet50", None, 128, 0.1)
model.sm3_dtype = torch.float32
model.to(T.DEV)
tr = SM3Trainer(model, lr=3e-4, weight_decay=5e-2, eps=1e-5, style=0)
derm, clinic = T._latent_batch(B, 64, 7)
fresh = T._latent_batch(B, 64, 8)
for step in range(31):
    if step in (0, 3, 6, 10, 15, 20, 30):
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for name, (d, c) in (("train", (derm, clinic)), ("fresh", fresh)):
            res = {}
            for dt in (torch.float32, torch.bfloat16):
                m = T._build(0, dt, sd); m.train()
                res[dt] = T._compat_step(m, d, c)
                del m
            lf, lb = res[torch.float32][1], res[torch.bfloat16][1]
            dl = max(float((a - b).abs().max()) for a, b in zip(lf, lb))
            rms = max(float((a - b).pow(2).mean().sqrt()) for a, b in zip(lf, lb))
            gn = {dt: float(torch.sqrt(sum((g.double() ** 2).sum() for g in res[dt][2].values()))) for dt in res}
            num = sum(((res[torch.float32][2][k].double() - res[torch.bfloat16][2][k].double()) ** 2).sum() for k in res[torch.float32][2])
            print(f"step {step:2d} {name}: loss f32 {res[torch.float32][0]:.4f} bf16 {res[torch.bfloat16][0]:.4f} max|dlogit| {dl:.3f} rms {rms:.4f} "
                  f"|g| f32 {gn[torch.float32]:.4f} bf16 {gn[torch.bfloat16]:.4f} relL2(g) {float(num.sqrt())/gn[torch.float32]:.3f}", flush=True)
    tr.step(derm, clinic)
