import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from oracle import procedural, sm3_oracle as O
from src.models.simclr import SimCLRSkinV32

def run(B, size, seed, style):
    state = procedural.make_state_dict(seed=seed)
    derm_np, clinic_np = procedural.make_pair_batch(B, size, seed)
    P, Bf = O.split_state(state, torch.float64)
    derm = [torch.from_numpy(a).double() for a in derm_np]; clinic = [torch.from_numpy(a).double() for a in clinic_np]
    outs = O.sm3_v32_forward(P, Bf, derm, clinic, style, 0.1, True)
    loss = O.sm3_loss(outs, style); loss.backward()
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.sm3_dtype = torch.float32
    model.cuda().train()
    d = [torch.from_numpy(a).cuda() for a in derm_np]; c = [torch.from_numpy(a).cuda() for a in clinic_np]
    crit = torch.nn.CrossEntropyLoss()
    o = model(d, c, style)
    w = 0.25 if style == 2 else 0.5
    l = crit(*o[0]) + crit(*o[1]) + sum(w * crit(*x) for x in o[2])
    l.backward()
    print(f"B={B} size={size} style={style}: loss hip {float(l):.6f} oracle64 {float(loss):.6f}")
    print("  logits maxdiff derm %.2e clinic %.2e cross %s" % (
        (o[0][0].detach().cpu().double() - outs[0][0].detach()).abs().max(),
        (o[1][0].detach().cpu().double() - outs[1][0].detach()).abs().max(),
        [float((a[0].detach().cpu().double() - b[0].detach()).abs().max()) for a, b in zip(o[2], outs[2])]))
    groups = {}
    for k, p in model.named_parameters():
        gk = ".".join(k.split(".")[:2]) if not k.startswith("cross") else ".".join(k.split(".")[:2])
        if "encoder" in k:
            gk = ".".join(k.split(".")[:3])
        a = groups.setdefault(gk, [0.0, 0.0])
        a[0] += float(p.grad.double().pow(2).sum()); a[1] += float(P[k].grad.pow(2).sum())
    for gk, (a, b) in groups.items():
        print(f"  {gk:45s} |g| hip {a**0.5:.5e} oracle {b**0.5:.5e} ratio {(a/b)**0.5:.4f}")

run(3, 96, 2, 2)
run(3, 96, 2, 0)
run(4, 96, 2, 2)
