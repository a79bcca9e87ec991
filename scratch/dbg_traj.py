import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from oracle import procedural, sm3_oracle as O
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
seed, batch, size, lr = 11, 4, 64, float(os.environ.get("LR", "1e-4"))
state = procedural.make_state_dict(seed=seed)
P, B = O.split_state(state, torch.float64)
p0 = {k: v.detach().clone() for k, v in P.items()}
model = SimCLRSkinV32("resnet50", None, 128, 0.1)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
model.sm3_dtype = torch.float32; model.to("cuda:0")
tr = SM3Trainer(model, lr=lr, weight_decay=5e-2, eps=1e-5, style=0)
opt = {}
derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
want, _ = O.train_step(P, B, [torch.from_numpy(a).double() for a in derm_np], [torch.from_numpy(a).double() for a in clinic_np], 0, 0.1, opt, lr)
got = tr.step([torch.from_numpy(a).cuda() for a in derm_np], [torch.from_numpy(a).cuda() for a in clinic_np])
torch.cuda.synchronize()
print("loss", float(got), float(want))
sd = model.state_dict()
names = [n for n, _ in model.named_parameters()]
gv = dict(zip(names, tr._engine().store.grad_views()))
rows = []
for k in names:
    d_or = (P[k].detach() - p0[k]).reshape(-1)
    d_hip = (sd[k].double().cpu() - p0[k]).reshape(-1)
    cos = float((d_or @ d_hip) / (d_or.norm() * d_hip.norm() + 1e-300))
    g_or = P[k].grad.reshape(-1); g_hip = gv[k].double().cpu().reshape(-1)
    gcos = float((g_or @ g_hip) / (g_or.norm() * g_hip.norm() + 1e-300))
    flips = float((torch.sign(g_or) != torch.sign(g_hip)).double().mean())
    rows.append((cos, k, float(d_or.norm()), float(d_hip.norm()), gcos, flips, float(g_or.abs().median())))
rows.sort()
for r in rows[:25]:
    print("cos %.4f %-55s |d_or| %.3e |d_hip| %.3e gcos %.5f signflips %.4f med|g| %.2e" % r)
print("mean cos", np.mean([r[0] for r in rows]), "mean flips", np.mean([r[5] for r in rows]))
