"""How tight can train-mode (batch-statistics) encoder gradients be checked against the fp64 oracle?  (experiment)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import procedural, sm3_oracle as O
import resnet
for B, S in ((16, 64), (8, 128)):
    state = procedural.make_state_dict(procedural.resnet50_spec(""), seed=23)
    x = torch.from_numpy(procedural.make_images(B, S, 23, "derm0"))
    P, Bf = O.split_state(state, torch.float64)
    f_ref = O.resnet50_features(x.double(), P, Bf, "", True)
    (f_ref ** 2).sum().backward()
    P32, Bf32 = O.split_state(state, torch.float32)
    f32 = O.resnet50_features(x.float(), P32, Bf32, "", True)
    (f32 ** 2).sum().backward()
    m = resnet.resnet50(weights=None); m.fc = torch.nn.Identity()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = torch.float32
    m.cuda().train()
    f = m(x.cuda()); (f.double() ** 2).sum().backward(); torch.cuda.synchronize()
    errs, errs32 = [], []
    for k, p in m.named_parameters():
        r = P[k].grad
        errs.append((float((p.grad.double().cpu() - r).norm() / (r.norm() + 1e-30)), k))
        errs32.append((float((P32[k].grad.double() - r).norm() / (r.norm() + 1e-30)), k))
    errs.sort(); errs32.sort()
    print(B, S, "feat rel", float((f.detach().cpu().double() - f_ref.detach()).norm() / f_ref.detach().norm()),
          "HIP f32 vs fp64: median", errs[len(errs) // 2][0], "max", errs[-1], "| torch fp32 vs fp64: median", errs32[len(errs32) // 2][0], "max", errs32[-1])
