"""Round 5: spread of the T2 linear-probe AUROC (tests/test_round3_gpu.py::test_T2_linear_probe_auroc_after_stream_training)
over repeated runs per arithmetic mode and over the size of the held-out set -- the numbers its bounds are derived from.
usage (GPU box, repo root): python scratch/r5_auroc_spread.py [runs_per_mode]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_round3_gpu as T  # noqa: E402
from sm3hip.trainer import SM3Trainer  # noqa: E402
from src.models.simclr import SimCLRSkinV32  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S, nb, B, steps = 64, 16, 64, 128
torch.manual_seed(5)
init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
tr_d, tr_c, tr_y = T._latent_set(2048, S, 7, dc=1.0)
tests = {n: T._latent_set(n, S, 8, dc=1.0) for n in (2048, 8192)}
train_set = (tr_d[0], tr_c[0], tr_y)
stream = [T._latent_set(B, S, 100 + i, views=2, dc=1.0)[:2] for i in range(nb)]
m0 = T._build(0, torch.float32, init)
print("untrained", {n: round(T._probe_auroc(m0, train_set, (t[0][0], t[1][0], t[2])), 4) for n, t in tests.items()}, flush=True)
del m0
for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
    for r in range(runs):
        model = T._build(0, dt, init)
        tr = SM3Trainer(model, lr=1e-3, weight_decay=5e-2, eps=1e-5, style=0, init_scale=1024.0)
        losses = [float(tr.step(*stream[s % nb])) for s in range(steps)]
        torch.cuda.synchronize()
        a = {n: round(T._probe_auroc(model, train_set, (t[0][0], t[1][0], t[2])), 4) for n, t in tests.items()}
        print(name, r, a, "final loss", round(float(np.mean(losses[-nb:])), 3), flush=True)
        del tr, model
        torch.cuda.empty_cache()
