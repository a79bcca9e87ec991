"""T2 probe, longer SSL training: does the frozen-feature AUROC of the label heads rise above the
untrained encoder's with enough steps on a stream of learnable batches, and how far apart do the arithmetic modes end?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scratch")]
import torch
import test_config_gpu as T
import test_round3_gpu as R3
from sm3hip.metrics import NUM_CLASSES, auc_avg
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
DEV, S = T.DEV, 64
torch.manual_seed(5)
init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
NTR, NTE = 2048, 2048
dtr, ctr, ytr = R3._labelled_latent(NTR, S, 7)
dte, cte, yte = R3._labelled_latent(NTE, S, 8)

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

def run(nb, B, checkpoints, lr, modes):
    stream = [T._latent_batch(B, S, 100 + i) for i in range(nb)]
    for name, dt in modes:
        model = T._build(0, dt, init)
        tr = SM3Trainer(model, lr=lr, weight_decay=5e-2, eps=1e-5, style=0, init_scale=1024.0)
        row = f"nb={nb} B={B} lr={lr} {name}:"
        s = 0
        for cp in checkpoints:
            while s < cp:
                loss = float(tr.step(*stream[s % nb])); s += 1
            row += f"  [{cp}] loss {loss:.3f} AUROC {probe(model):.4f}"
        print(row, flush=True)
        del tr, model
        torch.cuda.empty_cache()

M = (("f32", torch.float32), ("f32b", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16))
run(32, 64, (100, 200, 400, 800), 1e-3, M)
run(32, 64, (200, 400, 800), 3e-3, M[:1] + M[3:])
