"""smoke(): one tiny SM3 pre-training step on cuda:0 through the HIP path, checked against the CPU oracle
(the oracle is the checker here, never the thing shipped)."""
import os
import sys

import torch


def smoke():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import procedural, sm3_oracle as O
    from sm3hip import _lib
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32

    _lib.load()  # fail loudly if the HIP library is missing
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    B, size, seed = 4, 64, 1
    state = procedural.make_state_dict(seed=seed)
    derm_np, clinic_np = procedural.make_pair_batch(B, size, seed)

    # oracle (CPU fp32)
    P, Bf = O.split_state(state, torch.float32)
    derm = [torch.from_numpy(a) for a in derm_np]
    clinic = [torch.from_numpy(a) for a in clinic_np]
    ref_loss, _ = O.train_step(P, Bf, derm, clinic, 0, 0.1)

    # HIP path, exact-f32 MFMA mode
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.sm3_dtype = torch.float32
    model.to(dev)
    trainer = SM3Trainer(model, lr=1e-3)
    loss = trainer.step([d.to(dev) for d in derm], [c.to(dev) for c in clinic])
    torch.cuda.synchronize()
    got = float(loss)
    assert abs(got - float(ref_loss)) < 1e-3, (got, float(ref_loss))
    gn_ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in P.values()))
    eng = trainer._engine()
    gn = eng.store.flat_g.double().norm().cpu()
    assert abs(float(gn) - float(gn_ref)) < 5e-2 * float(gn_ref), (float(gn), float(gn_ref))
    print(f"smoke ok: loss {got:.6f} (oracle {float(ref_loss):.6f}), |grad| {float(gn):.5f} (oracle {float(gn_ref):.5f})")
