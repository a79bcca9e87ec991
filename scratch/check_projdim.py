import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from oracle import procedural, sm3_oracle as O
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
for proj_dim, temp, style in ((256, 0.5, 0), (64, 0.07, 2)):
    torch.manual_seed(1)
    model = SimCLRSkinV32("resnet50", None, proj_dim, temp)
    state = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    P, B = O.split_state(state, torch.float64)
    model.sm3_dtype = torch.float32; model.to("cuda:0")
    derm_np, clinic_np = procedural.make_pair_batch(8, 64, 9)
    want, _ = O.train_step(P, B, [torch.from_numpy(a).double() for a in derm_np], [torch.from_numpy(a).double() for a in clinic_np], style, temp)
    tr = SM3Trainer(model, lr=1e-6, style=style)
    got = tr.step([torch.from_numpy(a).cuda() for a in derm_np], [torch.from_numpy(a).cuda() for a in clinic_np])
    print("proj_dim", proj_dim, "T", temp, "style", style, "hip", float(got), "oracle", float(want))
