"""T2 probe with a stronger label signal: every sample's latent carries a per-channel offset (dc) that the label heads are
(mostly) functions of; SSL stream and probe sets share the distribution.  f32 x2 / fp16 / bf16 from one initialisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
import torch
import test_config_gpu as T
from sm3hip.metrics import NUM_CLASSES, auc_avg
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
DEV, S = T.DEV, 64
DC = float(os.environ.get("DC", "1.0"))
MIX = torch.tensor([[0.6, 0.3, 0.1], [0.2, 0.5, 0.3], [0.1, 0.2, 0.7]], device=DEV)

def latents(n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    z = torch.randn(n, 3, 6, 6, device=DEV, generator=g) + DC * torch.randn(n, 3, 1, 1, device=DEV, generator=g)
    return z, g

def render(z, g, views):
    base = torch.nn.functional.interpolate(z, size=(S, S), mode="bilinear", align_corners=False) * 1.5
    other = torch.einsum("dc,bchw->bdhw", MIX, base).flip(-1)
    noise = lambda t: (t + 0.5 * torch.randn(t.shape, device=DEV, generator=g)).contiguous()
    return [noise(base) for _ in range(views)], [noise(other) for _ in range(views)]

def labels(z):
    stats = torch.cat([z.mean((2, 3)), z.abs().mean((2, 3)), (z ** 2).mean((1, 2, 3)).unsqueeze(1)], 1)
    stats = (stats - stats.mean(0)) / stats.std(0)
    R = torch.randn(7, 8, generator=torch.Generator().manual_seed(1234)).to(DEV)
    score = stats @ R
    out = []
    for i, nc in enumerate(NUM_CLASSES):
        qs = torch.quantile(score[:, i], torch.linspace(0, 1, nc + 1, device=DEV)[1:-1])
        out.append(torch.bucketize(score[:, i].contiguous(), qs))
    return torch.stack(out, 1)

torch.manual_seed(5)
init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
NTR = NTE = 2048
ztr, gtr = latents(NTR, 7); (dtr,), (ctr,) = render(ztr, gtr, 1); ytr = labels(ztr)
zte, gte = latents(NTE, 8); (dte,), (cte,) = render(zte, gte, 1); yte = labels(zte)

def probe(model):
    model.eval()
    with torch.no_grad():
        ftr = torch.cat([torch.cat(model.extract(dtr[i:i + 512], ctr[i:i + 512]), 1) for i in range(0, NTR, 512)]).double()
        fte = torch.cat([torch.cat(model.extract(dte[i:i + 512], cte[i:i + 512]), 1) for i in range(0, NTE, 512)]).double()
    mu, sd = ftr.mean(0), ftr.std(0) + 1e-6
    Xtr = torch.cat([(ftr - mu) / sd, torch.ones(len(ftr), 1, dtype=torch.float64, device=DEV)], 1)
    Xte = torch.cat([(fte - mu) / sd, torch.ones(len(fte), 1, dtype=torch.float64, device=DEV)], 1)
    A = Xtr.t() @ Xtr + 200.0 * torch.eye(Xtr.shape[1], dtype=torch.float64, device=DEV)
    preds = []
    for i, nc in enumerate(NUM_CLASSES):
        Y = torch.nn.functional.one_hot(ytr[:, i], nc).double()
        preds.append(Xte @ torch.linalg.solve(A, Xtr.t() @ Y))
    model.train()
    return float(auc_avg(preds, yte)[1])

m0 = T._build(0, torch.float32, init)
print(f"DC={DC} untrained: AUROC {probe(m0):.4f}", flush=True)
del m0
nb, B, lr = 16, 64, 1e-3
stream = []
for i in range(nb):
    z, g = latents(B, 100 + i)
    stream.append(render(z, g, 2))
for name, dt in (("f32", torch.float32), ("f32b", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
    model = T._build(0, dt, init)
    tr = SM3Trainer(model, lr=lr, weight_decay=5e-2, eps=1e-5, style=0, init_scale=1024.0)
    row = f"DC={DC} nb={nb} B={B} lr={lr} {name}:"
    s = 0
    for cp in (32, 64, 128):
        while s < cp:
            loss = float(tr.step(*stream[s % nb])); s += 1
        row += f"  [{cp}] loss {loss:.3f} AUROC {probe(model):.4f}"
    print(row, flush=True)
    del tr, model
    torch.cuda.empty_cache()
