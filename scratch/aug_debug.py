import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_augment as T
from oracle import augment_oracle as A
from sm3hip.augment import SimCLRAugment
B, Hs, Ws, size = 4, 150, 210, (64, 64)
src = T._src(B, Hs, Ws, 7)
aug = SimCLRAugment(size, T.MEAN, T.STD)
g = torch.Generator().manual_seed(11)
base = aug.sample(B, Hs, Ws, g)
def run(tag, mod):
    p = aug.sample(B, Hs, Ws, torch.Generator().manual_seed(11))
    p.ops[:] = 0; p.gray[:] = 0; p.flip[:] = 0; p.sigma[:] = 0
    mod(p)
    out = aug.apply(src.cuda(), p).cpu().double()
    errs = []
    for b in range(B):
        ref = A.augment_one(src[b], p.box[b], bool(p.flip[b]), p.ops[:, b], p.factors[:, b], bool(p.gray[b]), float(p.sigma[b]), T.MEAN, T.STD, *size)
        errs.append(float((out[b] - ref).abs().max()))
    print(tag, ["%.2e" % e for e in errs], flush=True)
run("crop only", lambda p: None)
run("flip", lambda p: p.flip.fill_(1))
for op in (1, 2, 3, 4):
    def m(p, op=op):
        p.ops[0, :] = op; p.factors[0, :] = torch.tensor([0.17, 1.6, 0.3, 0.05]) if op != 4 else torch.tensor([0.17, -0.1, 0.05, 0.2])
    run(f"op {op}", m)
run("gray", lambda p: p.gray.fill_(1))
run("blur", lambda p: p.sigma.copy_(torch.tensor([0.1, 0.7, 1.3, 2.0])))
