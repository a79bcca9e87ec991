"""Linear-probe row: oracle pinned to the reference's Baseline golden, AUROC aggregation against
sklearn, module surface (CPU); HIP path parity (GPU)."""
import os

import numpy as np
import pytest
import torch


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "baseline_b6_s64_f64.npz"))


def _inputs(g):
    from oracle import procedural
    batch, size, seed = [int(v) for v in g["meta"]]
    state = procedural.make_state_dict(procedural.baseline_spec(), seed=seed)
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    return state, torch.from_numpy(derm_np[0]), torch.from_numpy(clinic_np[0]), torch.from_numpy(g["labels"]).long()


def _sub(t, n=512):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].double().cpu().numpy()


def test_oracle_baseline_matches_reference_golden(golden_dir):
    from oracle import sm3_oracle as O
    g = _golden(golden_dir)
    state, derm, clinic, labels = _inputs(g)
    P, B = O.split_state(state, torch.float64)
    outs = O.baseline_forward(P, B, derm.double(), clinic.double(), training=False)
    loss = O.linear_probe_loss(outs, labels, tuple(g["label_weights"]))
    loss.backward()
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f"logits_{i}"], atol=1e-9)
        np.testing.assert_allclose(_sub(P[f"classifier.{i}.weight"].grad), g[f"grad_w_{i}"], atol=1e-10)
        np.testing.assert_allclose(P[f"classifier.{i}.bias"].grad.numpy(), g[f"grad_b_{i}"], atol=1e-10)
    assert abs(float(loss) - float(g["loss"])) < 1e-9


def test_auroc_matches_sklearn_and_reference_aggregation():
    from oracle import sm3_oracle as O
    from sm3hip.metrics import NUM_CLASSES, auc_avg, multiclass_auroc
    from sklearn.metrics import roc_auc_score
    g = torch.Generator().manual_seed(1)
    n = 300
    preds = [(torch.randn(n, c, generator=g) * 2).round() / 2 for c in NUM_CLASSES]  # rounded: plenty of ties
    targets = torch.stack([torch.randint(0, c, (n,), generator=g) for c in NUM_CLASSES], dim=1)
    for i, c in enumerate(NUM_CLASSES):
        got = multiclass_auroc(preds[i], targets[:, i], c).numpy()
        prob = torch.softmax(preds[i].double(), 1).numpy()
        want = np.array([roc_auc_score((targets[:, i].numpy() == k).astype(int), prob[:, k]) for k in range(c)])
        np.testing.assert_allclose(got, want, atol=1e-12)
    per, avg = auc_avg(preds, targets)
    per_ref, avg_ref = O.auroc_selected(preds, targets)
    np.testing.assert_allclose([float(v) for v in per], per_ref, atol=1e-12)
    assert abs(float(avg) - avg_ref) < 1e-12
    # a class that never occurs: 0, as torchmetrics reports it
    assert float(multiclass_auroc(preds[0], torch.zeros(n, dtype=torch.long), 5)[2]) == 0.0


def test_baseline_module_surface(golden_dir):
    from src.models.baseline import Baseline
    keys = open(os.path.join(golden_dir, "baseline_state_dict_keys.txt")).read().split()
    m = Baseline("resnet50", None)
    assert list(m.state_dict().keys()) == keys
    m.freeze_backbone()
    assert sum(p.requires_grad for p in m.parameters()) == 16
    with pytest.raises(NotImplementedError):
        Baseline("resnet18")


@pytest.mark.gpu
def test_conv_bn_act_eval_kernel():
    import torch.nn.functional as F
    from sm3hip import ops
    D = torch.device("cuda:0")
    for dt in (torch.float32, torch.bfloat16):
        code = ops.dtype_code(dt)
        g = torch.Generator().manual_seed(3)
        N, Ci, Co, H, k, s = 3, 64, 192, 11, 3, 2
        x = torch.randn(N, Ci, H, H, generator=g).to(dt).float()
        w = (torch.randn(Co, Ci, k, k, generator=g) / 24).to(dt).float()
        scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
        ref = F.conv2d(x.double(), w.double(), stride=s, padding=1)
        res = torch.randn(ref.shape, generator=g).to(dt).float()
        want = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
        d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, 1)
        y = torch.empty(N, d.Ho, d.Wo, Co, dtype=dt, device=D)
        ops.conv_bn_act_eval(d, x.permute(0, 2, 3, 1).contiguous().to(dt).to(D),
                             w.permute(0, 2, 3, 1).contiguous().to(dt).to(D), scale.to(D), shift.to(D),
                             res.permute(0, 2, 3, 1).contiguous().to(dt).to(D), True, y)
        torch.cuda.synchronize()
        got = y.float().cpu().permute(0, 3, 1, 2).double()
        tol = (2e-5 if dt == torch.float32 else 1.2e-2) * float(want.abs().max())
        assert float((got - want).abs().max()) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("with_res", [True, False], ids=["general_epilogue", "lean_epilogue"])
def test_conv_bn_eval_derives_scale_shift_in_the_epilogue(with_res, affine):
    """sm3_conv_bn_eval (BatchNorm tensors handed to the launch) == sm3_bn_eval_scale_shift + sm3_conv_bn_act_eval, bit for
    bit, in every dtype; affine=False is the projector's last BatchNorm1d."""
    from sm3hip import ops
    D = torch.device("cuda:0")
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        code = ops.dtype_code(dt)
        g = torch.Generator().manual_seed(5)
        N, Ci, Co, H, k, s = 3, 128, 200, 9, 3, 1
        x = torch.randn(N, H, H, Ci, generator=g).to(dt).to(D)
        w = (torch.randn(Co, k, k, Ci, generator=g) / 34).to(dt).to(D)
        gamma = (torch.rand(Co, generator=g) + 0.5).to(D) if affine else None
        beta = torch.randn(Co, generator=g).to(D) if affine else None
        rm, rv = torch.randn(Co, generator=g).to(D), (torch.rand(Co, generator=g) + 0.1).to(D)
        d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, 1)
        res = torch.randn(N * d.Ho * d.Wo, Co, generator=g).to(dt).to(D) if with_res else None
        scale, shift = torch.empty(Co, device=D), torch.empty(Co, device=D)
        ops.bn_eval_scale_shift(gamma, beta, rm, rv, 1e-5, Co, scale, shift)
        y0 = torch.empty(N * d.Ho * d.Wo, Co, dtype=dt, device=D)
        ops.conv_bn_act_eval(d, x, w, scale, shift, res, True, y0)
        y1 = torch.full_like(y0, float("nan"))
        ops.conv_bn_eval(d, x, w, gamma, beta, rm, rv, 1e-5, res, True, y1)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)


@pytest.mark.gpu
def test_baseline_linear_probe_step_matches_golden(golden_dir):
    """Frozen eval-mode encoders on the fused inference kernels + the 8 heads: logits, weighted loss and head
    gradients of the reference (tools/backbone_eval.py:98-112 with --finetune fc)."""
    from src.models.baseline import Baseline
    g = _golden(golden_dir)
    state, derm, clinic, labels = _inputs(g)
    m = Baseline("resnet50", None)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.freeze_backbone()
    for bb in (m.derm_backbone, m.clinic_backbone):
        bb.sm3_dtype = torch.float32
    m.cuda().eval()
    outs = m([derm.cuda(), clinic.cuda()])
    crit = torch.nn.CrossEntropyLoss()
    loss = sum(float(w) * crit(o, labels[:, i].cuda()) for i, (o, w) in enumerate(zip(outs, g["label_weights"]))) / 8
    loss.backward()
    torch.cuda.synchronize()
    scale = max(float(np.abs(g[f"logits_{i}"]).max()) for i in range(8))
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().cpu().double().numpy(), g[f"logits_{i}"], atol=2e-4 * scale)
        gw = g[f"grad_w_{i}"]
        np.testing.assert_allclose(_sub(m.classifier[i].weight.grad), gw, atol=2e-4 * float(np.abs(gw).max()) + 1e-9)
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * float(g["loss"])
    assert int(m.derm_backbone.bn1.num_batches_tracked) == 0  # eval mode: running statistics untouched
    # bf16 inference path: same ranking of classes on almost every sample
    for bb in (m.derm_backbone, m.clinic_backbone):
        bb.sm3_dtype = torch.bfloat16
    with torch.no_grad():
        outs16 = m([derm.cuda(), clinic.cuda()])
    rel = max(float((a.double().cpu() - torch.from_numpy(g[f"logits_{i}"])).norm() / np.linalg.norm(g[f"logits_{i}"]))
              for i, a in enumerate(outs16))
    assert rel < 3e-2, rel
