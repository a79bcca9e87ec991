"""Pins the CPU oracle (oracle/sm3_oracle.py) against fixtures produced by the reference's own
Python (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import procedural, sm3_oracle as O

CASES = [
    ("b4_s64_f32", torch.float32, 2e-4, 2e-3),
    ("b4_s64_f64", torch.float64, 1e-9, 1e-8),
    ("b3_s96_style2_f64", torch.float64, 1e-9, 1e-8),
    ("b8_s64_style1_f64", torch.float64, 1e-9, 1e-8),
    ("b32_s64_f64", torch.float64, 1e-9, 1e-8),  # round 2: the well-conditioned case (extract features: first 8 rows)
]


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, f"sm3_v32_{tag}.npz"))


def _sub(t, n=256):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].double().numpy()


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def stepped(request, golden_dir):
    tag, dtype, rtol, logit_atol = request.param
    g = _load(golden_dir, tag)
    batch, size, seed, style = [int(v) for v in g["meta"]]
    torch.set_num_threads(8)
    P, B = O.split_state(procedural.make_state_dict(seed=seed), dtype)
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    derm = [torch.from_numpy(a).to(dtype) for a in derm_np]
    clinic = [torch.from_numpy(a).to(dtype) for a in clinic_np]
    taps = {}
    for p in P.values():
        p.grad = None
    outs = O.sm3_v32_forward(P, B, derm, clinic, style, 0.1, True, None, taps)
    loss = O.sm3_loss(outs, style)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in P.items()}
    with torch.no_grad():
        feats = O.extract(P, B, derm[0], clinic[0])  # pre-step weights, post-forward running stats
    opt = {"step": 1}
    with torch.no_grad():
        for k, p in P.items():
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            O.adamw_step(p, p.grad, m, v, 1, float(g["lr"]), eps=1e-5, weight_decay=5e-2)
    return dict(g=g, P=P, B=B, outs=outs, loss=loss.detach(), grads=grads, taps=taps, rtol=rtol, feats=feats,
                logit_atol=logit_atol, derm=derm, clinic=clinic, style=style)


def test_logits_and_loss(stepped):
    g, outs = stepped["g"], stepped["outs"]
    a = stepped["logit_atol"]
    np.testing.assert_allclose(outs[0][0].detach().double().numpy(), g["derm_logits"], atol=a, rtol=0)
    np.testing.assert_allclose(outs[1][0].detach().double().numpy(), g["clinic_logits"], atol=a, rtol=0)
    for i, (lg, lab) in enumerate(outs[2]):
        np.testing.assert_allclose(lg.detach().double().numpy(), g[f"cross_logits_{i}"], atol=a, rtol=0)
        assert lab.dtype == torch.long and int(lab.abs().sum()) == 0
    assert abs(float(stepped["loss"]) - float(g["loss"])) < max(a, 1e-9)


def test_taps(stepped):
    g = stepped["g"]
    pre = "derm_backbone.encoder."
    names = {"conv1": pre + "conv1", "maxpool": pre + "maxpool"}
    for li, nb in enumerate((3, 4, 6, 3), start=1):
        for b in range(nb):
            names[f"layer{li}.{b}"] = f"{pre}layer{li}.{b}"
    for short, full in names.items():
        t = stepped["taps"][full]
        assert list(t.shape) == list(g["tap_shape." + short])
        scale = float(g["tap_stat." + short][1]) + 1e-12
        np.testing.assert_allclose(_sub(t), g["tap_sub." + short], atol=stepped["rtol"] * 10 * scale, rtol=0)


def test_grads(stepped, golden_dir):
    g = stepped["g"]
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    assert names == list(stepped["P"].keys())
    gn = np.array([stepped["grads"][k].double().norm().item() for k in names])
    tol = 5e-2 if stepped["rtol"] > 1e-6 else 1e-6   # fp32 rounding noise is amplified by BN(affine=False)+normalize
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=tol, atol=1e-7)
    for key in g.files:
        if key.startswith("grad_full."):
            k = key[len("grad_full."):]
            ref = g[key]
            np.testing.assert_allclose(stepped["grads"][k].double().numpy(), ref,
                                       atol=tol * np.abs(ref).max(), rtol=0)
        if key.startswith("grad_sub."):
            k = key[len("grad_sub."):]
            ref = g[key]
            np.testing.assert_allclose(_sub(stepped["grads"][k]), ref, atol=tol * np.abs(ref).max(), rtol=0)


def test_adamw_and_buffers(stepped, golden_dir):
    g, P, B = stepped["g"], stepped["P"], stepped["B"]
    names = list(P.keys())
    pn = np.array([P[k].detach().double().norm().item() for k in names])
    # step-1 Adam is g/(|g|+eps): sign-like, so fp32 rounding noise in tiny grads moves a few
    # params by O(lr) -- the fp64 cases pin the formula tightly, fp32 only loosely.
    np.testing.assert_allclose(pn, g["post_param_norm"], rtol=1e-3 if stepped["rtol"] > 1e-6 else 1e-9)
    keys = open(os.path.join(golden_dir, "state_dict_keys.txt")).read().split()
    bn_keys = [k for k in keys if k.endswith(("running_mean", "running_var"))]
    bs = np.array([B[k].double().norm().item() for k in bn_keys])
    np.testing.assert_allclose(bs, g["post_buf_norm"], rtol=1e-4 if stepped["rtol"] > 1e-6 else 1e-9)
    nbt = np.array([int(B[k]) for k in keys if k.endswith("num_batches_tracked")])
    np.testing.assert_array_equal(nbt, g["post_nbt"])
    for key in g.files:
        if key.startswith("post_buf_full."):
            k = key[len("post_buf_full."):]
            np.testing.assert_allclose(B[k].double().numpy(), g[key], rtol=1e-4 if stepped["rtol"] > 1e-6 else 1e-9,
                                       atol=1e-5 if stepped["rtol"] > 1e-6 else 1e-12)


def test_extract_eval(stepped):
    g, P, B = stepped["g"], stepped["P"], stepped["B"]
    fd, fc = stepped["feats"]
    tol = 2e-3 if stepped["rtol"] > 1e-6 else 1e-8
    n = g["extract_derm"].shape[0]  # the B = 32 fixture stores the first 8 rows only
    np.testing.assert_allclose(fd[:n].double().numpy(), g["extract_derm"], rtol=tol, atol=tol)
    np.testing.assert_allclose(fc[:n].double().numpy(), g["extract_clinic"], rtol=tol, atol=tol)


def test_closed_form_equals_logits_ce():
    torch.manual_seed(0)
    z = torch.randn(16, 128, dtype=torch.float64)
    lg, lab = O.ntxent_logits(z, 0.1)
    ce = torch.nn.functional.cross_entropy(lg, lab)
    assert abs(float(ce) - float(O.ntxent_loss_closed_form(z, 0.1))) < 1e-12
    assert abs(float(ce) - float(O.cross_entropy_zero_label(lg))) < 1e-12


def test_state_dict_layout(golden_dir):
    keys = open(os.path.join(golden_dir, "state_dict_keys.txt")).read().split()
    assert len(keys) == 700
    spec = procedural.sm3_v32_spec()
    assert [k for k, _ in spec] == keys
    n_params = sum(int(np.prod(s)) for k, s in spec
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 81651840
