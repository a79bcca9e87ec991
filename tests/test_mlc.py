"""f-2 second half: training of the multi-label heads (tools/mlc_train.py) on the HIP kernels (sm3hip/mlc.py,
csrc/heads_train.hip) against stock PyTorch autograd of the reference's own module structure, and the spherical k-means
against a restatement of mlc_train.py:146-177."""
import math

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NUM_CLASSES = [5, 3, 2, 3, 3, 3, 3, 2]


class _RefHeads(nn.Module):
    """The head part of the reference's Model (mlc_train.py:58-90), stock PyTorch."""

    def __init__(self, in_dim, D, nhead, ff, dropout, l2_norm, bias):
        super().__init__()
        from src.models.projector import MultiLabelProjector4
        self.projectors = MultiLabelProjector4(in_dim, D, 8)
        self.mlc_sa = nn.TransformerEncoderLayer(d_model=D, nhead=nhead, dim_feedforward=ff, dropout=dropout)
        self.prototypes = nn.ModuleList([nn.Linear(D, n, bias=bias) for n in NUM_CLASSES])
        self.l2_norm = l2_norm

    def forward(self, feats):
        sa = self.mlc_sa(torch.stack(self.projectors(feats), dim=0))
        if self.l2_norm:
            sa = nn.functional.normalize(sa, dim=-1, p=2)
        return sa, [self.prototypes[i](sa[i % len(sa)]) for i in range(len(self.prototypes))]


@pytest.mark.parametrize("cfg", [dict(D=512, nhead=1, ff=128, l2=False, bias=False),   # run.sh:39-47 (mlc_train)
                                 dict(D=256, nhead=4, ff=256, l2=True, bias=True)],    # mlc_eval-style heads, parser defaults
                         ids=["v4_512_h1", "256_h4_l2_bias"])
def test_heads_training_matches_torch_autograd(cfg):
    from sm3hip import mlc
    torch.manual_seed(3)
    B, in_dim, T = 24, 256, 0.7
    ref = _RefHeads(in_dim, cfg["D"], cfg["nhead"], cfg["ff"], 0.0, cfg["l2"], cfg["bias"]).double()
    ref.train()
    feats = torch.randn(B, in_dim, dtype=torch.float64, requires_grad=True)
    targets = torch.stack([torch.randint(0, n, (B,)) for n in NUM_CLASSES])
    crit = nn.CrossEntropyLoss()
    sa_ref, preds_ref = ref(feats)
    loss_ref = sum(crit(p / T, t) for p, t in zip(preds_ref, targets)) / 8     # mlc_train.py:252-261
    loss_ref.backward()

    hip = _RefHeads(in_dim, cfg["D"], cfg["nhead"], cfg["ff"], 0.0, cfg["l2"], cfg["bias"])
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip.to(DEV).train()
    f = feats.detach().float().to(DEV).requires_grad_(True)
    sa, preds = mlc.heads_forward(hip, f, seed=1)
    loss = sum(crit(p / T, t.to(DEV)) for p, t in zip(preds, targets)) / 8
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert float((sa.cpu().double() - sa_ref.detach()).abs().max()) < 1e-4
    for (k, p_ref), (_, p) in zip(ref.named_parameters(), hip.named_parameters()):
        err = float((p.grad.cpu().double() - p_ref.grad).norm() / (p_ref.grad.norm() + 1e-12))
        assert err < 2e-4, (k, err)
    err = float((f.grad.cpu().double() - feats.grad).norm() / feats.grad.norm())
    assert err < 2e-4, err
    # the fused pseudo-label cross-entropy kernel gives the same loss and d(logits)
    lk, dl = mlc.pseudo_label_loss([p.detach() for p in preds], targets.to(DEV), T)
    assert abs(float(lk) - float(loss_ref)) < 1e-5
    cat = torch.cat(preds, 1).detach().requires_grad_(True)
    l2 = sum(crit(p / T, t.to(DEV)) for p, t in zip(cat.split(NUM_CLASSES, 1), targets)) / 8
    l2.backward()
    assert float((dl - cat.grad).abs().max()) < 1e-7


def test_dropout_is_deterministic_unbiased_and_used_in_backward():
    from sm3hip import mlc
    torch.manual_seed(0)
    B, in_dim = 64, 128
    hip = _RefHeads(in_dim, 256, 2, 128, 0.1, False, False).to(DEV).train()
    f = torch.randn(B, in_dim, device=DEV)
    sa1, p1 = mlc.heads_forward(hip, f, seed=5)
    sa2, p2 = mlc.heads_forward(hip, f, seed=5)
    sa3, p3 = mlc.heads_forward(hip, f, seed=6)
    assert torch.equal(sa1, sa2) and not torch.equal(sa1, sa3)        # the mask is a pure function of (seed, element)
    hip.eval()
    sa_e, _ = mlc.heads_forward(hip, f, seed=5)                        # eval: dropout off (mlc_train.py:init_memory)
    hip.train()
    mean = torch.stack([mlc.heads_forward(hip, f, seed=100 + i)[0] for i in range(48)]).mean(0)
    assert float((mean - sa_e).abs().mean()) < 0.2 * float(sa_e.abs().mean())  # unbiased in expectation (LayerNorm is non-linear: loose)
    # finite-difference check of one weight through the dropped network (same seed = same mask)
    w = hip.mlc_sa.linear2.weight
    tgt = torch.stack([torch.randint(0, n, (B,), device=DEV) for n in NUM_CLASSES])
    crit = nn.CrossEntropyLoss()

    def loss_of():
        _, pr = mlc.heads_forward(hip, f, seed=9)
        return sum(crit(p, t) for p, t in zip(pr, tgt)) / 8
    loss = loss_of()
    hip.zero_grad()
    loss.backward()
    g = w.grad[3, 5].item()
    with torch.no_grad():
        w[3, 5] += 1e-2
        lp = float(loss_of())
        w[3, 5] -= 2e-2
        lm = float(loss_of())
        w[3, 5] += 1e-2
    assert abs((lp - lm) / 2e-2 - g) < 5e-3 + 0.05 * abs(g), ((lp - lm) / 2e-2, g)


def test_spherical_kmeans_matches_the_reference_algorithm():
    from sm3hip import mlc
    g = torch.Generator().manual_seed(4)
    N, D, K = 413, 512, 5                                    # derm7pt's training split is 413 cases
    centers = nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    emb = nn.functional.normalize(centers[torch.randint(0, K, (N,), generator=g)] + 0.3 * torch.randn(N, D, generator=g), dim=1)
    gk = torch.Generator().manual_seed(7)
    cent, assign = mlc.spherical_kmeans(emb.to(DEV), K, iters=10, generator=gk)
    # restatement of mlc_train.py:146-177 with the same initial centroids
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(7))[:K]
    c = emb[idx].double()
    e = emb.double()
    for it in range(11):
        a = (e @ c.t()).max(1).indices
        if it == 10:
            break
        for k in range(K):
            if (a == k).any():
                c[k] = e[a == k].sum(0) / (a == k).sum()
        c = nn.functional.normalize(c, dim=1)
    assert torch.equal(assign.cpu(), a)
    assert float((cent.cpu().double() - c).abs().max()) < 1e-5
    assert len(set(assign.cpu().tolist())) == K


def test_cluster_memory_reproduces_the_reference_function(golden_dir):
    """The tool's `cluster_memory` (HIP k-means kernels, prototype update) against the REFERENCE'S OWN function run on CPU
    (tools/mlc_train.py:116-189 -> oracle/gen_kmeans_golden.py -> tests/golden/mlc_kmeans_ref.npz): four memory banks (413
    cases x 512 / 256 / 128 dimensions, 96 x 128; 5 / 3 / 2 / 3 clusters, one with heavily overlapping clusters), the k-means
    seed the reference drew its initial centroids with.  Every memory index gets the reference's assignment; the centroids
    copied into the prototype layer agree to fp32 rounding."""
    import importlib.util
    import os
    import types
    import numpy as np
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")
    spec = importlib.util.spec_from_file_location("sm3_mlc_train", os.path.join(tools, "mlc_train.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    g = np.load(os.path.join(golden_dir, "mlc_kmeans_ref.npz"))
    for ci in range(4):
        N, D, K, kseed = [int(v) for v in g[f"c{ci}_meta"]]
        emb = torch.from_numpy(g[f"c{ci}_emb"]).to(DEV)
        index = torch.from_numpy(g[f"c{ci}_index"]).to(DEV)
        proto = nn.Linear(D, K, bias=False).to(DEV)
        args = types.SimpleNamespace(world_size=1, rank=0)
        with torch.no_grad():
            assign = mt.cluster_memory(args, proto, K, index, emb, generator=torch.Generator().manual_seed(kseed))
        torch.cuda.synchronize()
        assert torch.equal(assign.cpu(), torch.from_numpy(g[f"c{ci}_assign"])), ci
        assert float((proto.weight.detach().cpu() - torch.from_numpy(g[f"c{ci}_centroids"])).abs().max()) < 5e-6, ci


def test_mlc_train_tool_runs_and_learns(tmp_path):
    """tools/mlc_train.py end to end on synthetic data: frozen HIP extractor (eval mode), memory bank, per-epoch spherical
    k-means, pseudo-label training of the heads.  Every epoch's loss is finite, the heads move between the first and the last
    epoch while the frozen extractor does not, the checkpoint holds the reference's keys.  (The epoch-mean losses are NOT
    compared with each other: the pseudo-labels are re-clustered every epoch, mlc_train.py:116-189, so a cluster's index -- and
    with it the cross-entropy against the previous epoch's prototypes -- is permuted between epochs; a last-bit change of the
    extractor's features reorders the history, e.g. 0.999 / 3.212 / 1.195.  That the heads LEARN is pinned where labels are
    fixed: test_heads_training_matches_torch_autograd above.)"""
    import importlib.util
    import os
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")
    spec = importlib.util.spec_from_file_location("sm3_mlc_train", os.path.join(tools, "mlc_train.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    args = mt.get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-", "--epochs", "3", "-b", "32", "--num-samples", "96",
                                       "--img-sz", "64", "64", "--log-path", str(tmp_path), "--temperature", "1",
                                       "--mlc-proj-dim", "128", "--sa-dim-ff", "64", "--sa-dropout", "0.1", "-lr", "1e-3",
                                       "--save-freq", "1"])
    args.world_size = 1
    hist = mt.main(0, args)
    assert len(hist) == 3 and all(math.isfinite(v) and 0.0 < v < 20.0 for v in hist), hist
    first = torch.load(os.path.join(str(tmp_path), "ckp_0.pth"), map_location="cpu", weights_only=False)["state_dict"]
    last = torch.load(os.path.join(str(tmp_path), "ckp_2.pth"), map_location="cpu", weights_only=False)["state_dict"]
    moved = [k for k in first if k.startswith(("mlc_sa.", "prototypes.", "projectors.")) and first[k].is_floating_point()
             and not torch.equal(first[k], last[k])]
    frozen = [k for k in first if k.startswith("extractor.") and not torch.equal(first[k], last[k])]
    assert len(moved) > 10 and not frozen, (len(moved), frozen[:4])
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint.pth.tar"), map_location="cpu", weights_only=False)
    assert {"epoch", "state_dict", "optimizer"} <= set(ck)
    keys = list(ck["state_dict"].keys())
    assert any(k.startswith("extractor.derm_backbone.encoder.") for k in keys)
    assert "mlc_sa.self_attn.in_proj_weight" in keys and "prototypes.7.weight" in keys and "projectors.projectors.0.0.weight" in keys


def test_fc_mode_heads_run_without_dropout():
    """--finetune fc (reference tools/mlc_eval.py:124-128): mlc_sa is in eval mode, so the head forward is deterministic
    (no dropout mask, whatever the seed) and equals torch's own TransformerEncoderLayer in eval mode."""
    from sm3hip import mlc
    torch.manual_seed(2)
    D, ff = 128, 64
    ref = _RefHeads(2 * 2048, D, 1, ff, 0.1, False, True).to(DEV)
    feats = torch.randn(6, 2 * 2048, device=DEV)
    ref.train()
    ref.projectors.eval()
    ref.mlc_sa.eval()                                  # what tools/mlc_eval.py's set_train_modes(., "fc") leaves
    _, a = mlc.heads_forward(ref, feats, seed=1)
    _, b = mlc.heads_forward(ref, feats, seed=99)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    want = ref(feats)[1]
    for x, y in zip(a, want):
        assert float((x - y).abs().max()) < 2e-4 * (float(y.abs().max()) + 1.0)
    ref.mlc_sa.train()                                 # projector / all modes: dropout active, seed-dependent
    _, c = mlc.heads_forward(ref, feats, seed=1)
    _, d = mlc.heads_forward(ref, feats, seed=99)
    assert any(not torch.equal(x, y) for x, y in zip(c, d))


@pytest.mark.parametrize("mode", ["fc", "projector", "all"])
def test_mlc_eval_tool_finetunes_from_an_mlc_train_checkpoint(tmp_path, mode):
    """tools/mlc_eval.py: loads the mlc_train checkpoint format (bias-free prototypes dropped, strict=False), fine-tunes
    with real labels in the `projector` and `all` freeze modes (the latter sends gradients into the HIP encoders'
    layer1-4 through the autograd bridge), saves best_finetune.pth with the reference's keys."""
    import importlib.util
    import os
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")

    def load(name):
        spec = importlib.util.spec_from_file_location("sm3_" + name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    mt, me = load("mlc_train"), load("mlc_eval")
    targs = mt.get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-", "--epochs", "1", "-b", "16", "--num-samples", "32", "--img-sz", "64", "64",
                                        "--log-path", str(tmp_path / "train"), "--mlc-proj-dim", "128", "--sa-dim-ff", "64"])
    targs.world_size = 1
    mt.main(0, targs)
    hist = me.main(["--data-name", "synthetic", "--data-path", "-", "--epochs", "2", "-b", "16", "--steps-per-epoch", "3", "--val-steps", "2", "--img-sz", "64", "64",
                    "--log-path", str(tmp_path / "eval"), "--mlc-proj-dim", "128", "--sa-dim-ff", "64", "--finetune", mode,
                    "--pretrain-path", str(tmp_path / "train" / "ckp_0.pth")])
    assert len(hist) == 2 and all(math.isfinite(t["loss"]) and 0.0 <= v["AUC_AVG"] <= 1.0 for t, v in hist)
    ck = torch.load(str(tmp_path / "eval" / "best_finetune.pth"), map_location="cpu", weights_only=False)
    assert "prototypes.0.bias" in ck["state_dict"] and "extractor.clinic_backbone.encoder.layer4.2.bn3.weight" in ck["state_dict"]
