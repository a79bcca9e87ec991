"""More GPU parity cases against the CPU oracle (f32-MFMA mode, T0 tolerances of tests/test_e2e_gpu.py):
edge shapes (odd batch, non-square images, the 448x448 size of BASELINE config 5), SimCLRSkinV3's shared cross
projector, eval-mode forward, a bare ResNet-50 with autograd, SimCLR alone, fp16-style loss scaling in AdamW."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(seed, spec=None):
    from oracle import procedural
    return procedural.make_state_dict(spec, seed=seed)


def _imgs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: (torch.randn(B, 3, H, W, generator=g) * 0.7 + torch.randn(B, 3, 1, 1, generator=g))
    return [mk(), mk()], [mk(), mk()]


def _run_pair(cls_name, B, H, W, style, seed, cross):
    from oracle import sm3_oracle as O
    import src.models.simclr as M
    state = _state(seed)
    if cls_name == "SimCLRSkinV3":  # shared projector: rename cross_proj.0.* -> cross_proj.*, drop cross_proj.1.*
        state = {k.replace("cross_proj.0.", "cross_proj."): v for k, v in state.items() if not k.startswith("cross_proj.1.")}
    derm, clinic = _imgs(B, H, W, seed)
    P, Bf = O.split_state(state, torch.float64)
    outs = O.sm3_v32_forward(P, Bf, [d.double() for d in derm], [c.double() for c in clinic], style, 0.1, True,
                             cross=cross)
    loss = O.sm3_loss(outs, style)
    loss.backward()
    model = getattr(M, cls_name)("resnet50", None, 128, 0.1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model.sm3_dtype = torch.float32
    model.cuda().train()
    o = model([d.cuda() for d in derm], [c.cuda() for c in clinic], style)
    crit = torch.nn.CrossEntropyLoss()
    w = 0.25 if style == 2 else 0.5
    l = crit(*o[0]) + crit(*o[1]) + sum(w * crit(*x) for x in o[2])
    l.backward()
    torch.cuda.synchronize()
    assert abs(float(l.detach()) - float(loss.detach())) < 1e-3
    for got, ref in [(o[0], outs[0]), (o[1], outs[1])] + list(zip(o[2], outs[2])):
        assert (got[0].detach().cpu().double() - ref[0].detach()).abs().max().item() < 2e-3
    gn = {k: p.grad.double().norm().item() for k, p in model.named_parameters()}
    rn = {k: p.grad.norm().item() for k, p in P.items()}
    tot_g = np.sqrt(sum(v * v for v in gn.values()))
    tot_r = np.sqrt(sum(v * v for v in rn.values()))
    assert abs(tot_g - tot_r) < 3e-2 * tot_r
    bad = [k for k in gn if abs(gn[k] - rn[k]) > 6e-2 * rn[k] + 1e-6]
    assert len(bad) <= 3, bad[:5]


# B >= 4: a BatchNorm1d over 2-3 rows (B = 2 or 3) is singular by construction (test_e2e_gpu.py header)
@pytest.mark.parametrize("B,H,W,style", [(5, 64, 96, 1), (4, 448, 448, 0)], ids=["odd-batch-nonsquare", "448"])
def test_edge_shapes_v32(B, H, W, style):
    _run_pair("SimCLRSkinV32", B, H, W, style, seed=11, cross=("cross_proj.0.", "cross_proj.1."))


def test_v3_shared_cross_projector():
    _run_pair("SimCLRSkinV3", 6, 64, 64, 2, seed=12, cross=("cross_proj.", "cross_proj."))


def test_eval_mode_forward_uses_running_statistics():
    from oracle import sm3_oracle as O
    from src.models.simclr import SimCLRSkinV32
    state = _state(13)
    derm, clinic = _imgs(4, 64, 64, 13)
    P, Bf = O.split_state(state, torch.float64, requires_grad=False)
    outs = O.sm3_v32_forward(P, Bf, [d.double() for d in derm], [c.double() for c in clinic], 0, 0.1, False)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.sm3_dtype = torch.float32
    model.cuda().eval()
    with torch.no_grad():
        o = model([d.cuda() for d in derm], [c.cuda() for c in clinic], 0)
    for got, ref in [(o[0], outs[0]), (o[1], outs[1])] + list(zip(o[2], outs[2])):
        assert (got[0].cpu().double() - ref[0]).abs().max().item() < 2e-3
    sd = model.state_dict()
    assert int(sd["derm_backbone.encoder.bn1.num_batches_tracked"]) == 0  # eval: buffers untouched
    assert torch.equal(sd["derm_backbone.encoder.bn1.running_mean"].cpu(), torch.from_numpy(state["derm_backbone.encoder.bn1.running_mean"]))


def test_bare_resnet50_forward_backward():
    """resnet.__dict__['resnet50'](weights=None) as used by inference.py / Baseline: features + autograd."""
    from oracle import procedural, sm3_oracle as O
    import resnet
    spec = procedural.resnet50_spec("")
    state = procedural.make_state_dict(spec, seed=14)
    x = _imgs(3, 64, 64, 14)[0][0]
    P, Bf = O.split_state(state, torch.float64)
    f_ref = O.resnet50_features(x.double(), P, Bf, "", True)
    (f_ref ** 2).sum().backward()
    m = resnet.resnet50(weights=None)
    m.fc = torch.nn.Identity()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = torch.float32
    m.cuda().train()
    f = m(x.cuda())
    (f ** 2).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(f.detach().cpu().double(), f_ref.detach(), rtol=1e-3, atol=1e-3)
    for k in ("conv1.weight", "layer4.2.bn3.weight", "layer2.0.downsample.0.weight", "layer1.0.conv2.weight"):
        g, r = dict(m.named_parameters())[k].grad.cpu().double(), P[k].grad
        assert (g - r).norm() <= 6e-2 * r.norm(), k  # the fp32 noise floor of this depth at B=3 (cf. test_e2e_gpu.py)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_frozen_encoder_keeps_its_filter_banks_until_the_weights_really_change(dt):
    """Frozen-encoder loops skip the per-forward re-layout of the filter banks; the decision is a device-side hash of the
    fp32 masters, so a write that no version counter sees (`p.data`, the momentum-encoder idiom) is still picked up."""
    from oracle import procedural
    import resnet
    from sm3hip import ops, profiler
    state = procedural.make_state_dict(procedural.resnet50_spec(""), seed=19)

    def build():
        m = resnet.resnet50(weights=None)
        m.fc = torch.nn.Identity()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
        m.sm3_dtype = dt
        for p in m.parameters():
            p.requires_grad = False
        return m.cuda().eval()
    m = build()
    x = _imgs(4, 64, 64, 19)[0][0].cuda()
    with torch.no_grad():
        f = [m(x).clone() for _ in range(4)]          # calls 3 and 4 run on the kept banks
        assert all(torch.equal(f[0], fi) for fi in f[1:])
        m.layer3[1].conv2.weight.data.mul_(0.5)       # invisible to torch's version counters
        m.conv1.weight.data.add_(0.01)                # the direct stem's bank too
        g1 = m(x).clone()
        g2 = m(x).clone()
        ref = build()
        ref.layer3[1].conv2.weight.data.mul_(0.5)
        ref.conv1.weight.data.add_(0.01)
        g_ref = ref(x)
    torch.cuda.synchronize()
    assert not torch.equal(g1, f[0])
    assert torch.equal(g1, g_ref) and torch.equal(g2, g_ref)


def test_simclr_alone():
    from oracle import procedural, sm3_oracle as O
    from src.models.simclr import SimCLR
    spec = procedural.resnet50_spec("encoder.") + procedural.projector_spec("projector.")
    state = procedural.make_state_dict(spec, seed=15)
    x1, x2 = _imgs(4, 64, 64, 15)[0]
    P, Bf = O.split_state(state, torch.float64)
    (lg_ref, _), _ = O.simclr_forward(x1.double(), x2.double(), P, Bf, "", 0.5, True)
    m = SimCLR("resnet50", None, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = torch.float32
    m.cuda().train()
    lg, lab = m(x1.cuda(), x2.cuda())
    torch.cuda.synchronize()
    assert lab.dtype == torch.long and tuple(lg.shape) == (8, 7)
    assert (lg.detach().cpu().double() - lg_ref.detach()).abs().max().item() < 2e-3
    # return_feats=True standalone (reference simclr.py:54-91): the features of the SAME forward -- BatchNorm buffers
    # advance by exactly the two encoder calls of one forward, no hidden extra passes
    (lg_ref2, _), (f1_ref, f2_ref) = O.simclr_forward(x1.double(), x2.double(), P, Bf, "", 0.5, True)
    m2 = SimCLR("resnet50", None, 128, 0.5, return_feats=True)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m2.sm3_dtype = torch.float32
    m2.cuda().train()
    nbt0 = int(m2.encoder.bn1.num_batches_tracked)
    (lg2, lab2), (f1, f2) = m2(x1.cuda(), x2.cuda())
    torch.cuda.synchronize()
    assert int(m2.encoder.bn1.num_batches_tracked) == nbt0 + 2
    assert int(m2.encoder.layer4[2].bn3.num_batches_tracked) == nbt0 + 2
    assert torch.equal(lg2.detach(), lg.detach()) and not f1.requires_grad
    for got, ref in ((f1, f1_ref), (f2, f2_ref)):
        assert tuple(got.shape) == (4, 2048)
        assert torch.allclose(got.cpu().double(), ref.detach(), rtol=2e-3, atol=1e-3)  # f32 vs fp64 through 4-image BatchNorms


def test_adamw_grad_scale_and_overflow_skip_in_trainer_units():
    """GradScaler semantics folded into sm3_adamw: g*grad_scale, whole step skipped when found_inf is set."""
    from sm3hip import ops
    D = torch.device("cuda:0")
    p = torch.ones(1024, device=D); g = torch.full((1024,), 8.0, device=D)
    m, v = torch.zeros(1024, device=D), torch.zeros(1024, device=D)
    ops.adamw(p, g, m, v, 1e-2, 0.9, 0.999, 1e-5, 0.0, 1, grad_scale=1.0 / 8)
    torch.cuda.synchronize()
    assert torch.allclose(m, torch.full_like(m, 0.1)) and torch.allclose(p, torch.full_like(p, 1 - 1e-2), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("proj_dim,temp,style", [(256, 0.5, 0), (64, 0.07, 2)])
def test_other_projection_widths_temperatures_and_styles(proj_dim, temp, style):
    """Constructor arguments other than the recipe's (proj_dim 128, temperature 0.1): fused step loss against the
    CPU oracle in fp64 on the same weights and inputs (B = 8, 64x64, exact-f32 MFMA mode)."""
    from oracle import procedural, sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(1)
    model = SimCLRSkinV32("resnet50", None, proj_dim, temp)
    state = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    P, B = O.split_state(state, torch.float64)
    model.sm3_dtype = torch.float32
    model.to("cuda:0")
    derm_np, clinic_np = procedural.make_pair_batch(8, 64, 9)
    want, _ = O.train_step(P, B, [torch.from_numpy(a).double() for a in derm_np],
                           [torch.from_numpy(a).double() for a in clinic_np], style, temp)
    tr = SM3Trainer(model, lr=1e-6, style=style)
    got = tr.step([torch.from_numpy(a).cuda() for a in derm_np], [torch.from_numpy(a).cuda() for a in clinic_np])
    torch.cuda.synchronize()
    assert abs(float(got) - float(want)) < 1e-3, (float(got), float(want))
