"""End-to-end parity (GPU): the HIP engine behind the reference's module API against the golden vectors
generated from the reference itself (tests/golden, oracle/gen_golden.py) and against the CPU oracle.

Stated tolerances:
  T0  exact-f32 MFMA path vs fp64 goldens: logits atol 2e-3, loss atol 1e-3, features rtol 1e-3
      (cases b4_s64 and b8_s64).  The B=3 case (b3_s96_style2) is kept for the style-2 structure / logits
      layout only: a BatchNorm1d over 3 rows leaves post-ReLU channels with (near-)zero variance, so
      invstd = 1/sqrt(eps) = 316 amplifies fp32 rounding to ~6e-3 in the cross logits and ~6 % in the
      gradients that flow through cross_proj.0 (measured; B=4 at the same size is within 0.5 %).  Its bounds:
      logits atol 1.5e-2, loss 3e-3, gradient norms 12 %.
  T1  bf16 path vs f32 path: pooled features relative L2 <= 3e-2; the step-0 loss only within 1.0 (random-init
      conditioning: BatchNorm1d(affine=False) + L2-normalise + /0.1 amplifies rounding noise to the logit range;
      the reference's own bf16-autocast CPU run is off by 0.2-0.85)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, f"sm3_v32_{tag}.npz"))


def _sub(t, n=256):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].double().cpu().numpy()


def _build(seed, dtype):
    from oracle import procedural
    from src.models.simclr import SimCLRSkinV32
    state = procedural.make_state_dict(seed=seed)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model.sm3_dtype = dtype
    return model.to("cuda:0")


def _batch(batch, size, seed):
    from oracle import procedural
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    return ([torch.from_numpy(a).cuda() for a in derm_np], [torch.from_numpy(a).cuda() for a in clinic_np])


LOOSE = {"b3_s96_style2_f64"}


@pytest.fixture(scope="module", params=["b4_s64_f64", "b8_s64_style1_f64", "b3_s96_style2_f64"])
def compat_run(request, golden_dir):
    """Drop-in call contract: model(derm, clinic, style) -> logits; caller applies CrossEntropyLoss and
    backward(); torch.optim.AdamW steps (tools/backbone_train.py:98-127)."""
    g = _load(golden_dir, request.param)
    batch, size, seed, style = [int(v) for v in g["meta"]]
    model = _build(seed, torch.float32)
    model.train()
    derm, clinic = _batch(batch, size, seed)
    criterion = torch.nn.CrossEntropyLoss()
    opt = torch.optim.AdamW(model.parameters(), lr=float(g["lr"]), weight_decay=5e-2, eps=1e-5)
    outputs = model(derm, clinic, style)
    w = 0.25 if style == 2 else 0.5
    loss = criterion(*outputs[0]) + criterion(*outputs[1]) + sum(w * criterion(*o) for o in outputs[2])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    model.eval()
    with torch.no_grad():
        feats = model.extract(derm[0], clinic[0])
    model.train()
    opt.step()
    torch.cuda.synchronize()
    loose = request.param in LOOSE
    return dict(g=g, model=model, outputs=outputs, loss=float(loss.detach()), grads=grads, feats=feats, style=style,
                logit_atol=1.5e-2 if loose else 2e-3, loss_atol=3e-3 if loose else 1e-3,
                gn_rtol=0.12 if loose else 5e-2, l2_rtol=0.12 if loose else 6e-2)


def test_logits_loss_T0(compat_run):
    g, outs = compat_run["g"], compat_run["outputs"]
    assert outs[0][0].dtype == torch.float32 and outs[0][1].dtype == torch.long
    a = compat_run["logit_atol"]
    np.testing.assert_allclose(outs[0][0].detach().cpu().double().numpy(), g["derm_logits"], atol=a, rtol=0)
    np.testing.assert_allclose(outs[1][0].detach().cpu().double().numpy(), g["clinic_logits"], atol=a, rtol=0)
    assert len(outs[2]) == (4 if compat_run["style"] == 2 else 2)
    for i, (lg, lab) in enumerate(outs[2]):
        np.testing.assert_allclose(lg.detach().cpu().double().numpy(), g[f"cross_logits_{i}"], atol=a, rtol=0)
        assert int(lab.abs().sum()) == 0 and lab.shape[0] == lg.shape[0]
    assert abs(compat_run["loss"] - float(g["loss"])) < compat_run["loss_atol"]


def test_gradients(compat_run, golden_dir):
    g, grads = compat_run["g"], compat_run["grads"]
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    assert names == list(grads.keys())
    gn = np.array([grads[k].double().norm().item() for k in names])
    # same bound the fp32 oracle needs against the fp64 reference (tests/test_oracle_golden.py)
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=compat_run["gn_rtol"], atol=1e-7)
    # Element-wise: the reference's own fp32 run differs from its fp64 run by 2-3 % (relative L2, up to 12 % of
    # max on single elements) on this tiny, badly conditioned batch (B=4: BatchNorm1d over 4-8 rows feeding an
    # L2-normalise and a 1/0.1 temperature), so the bound here is that noise floor; tight backward checks live
    # in tests/test_kernels_gpu.py.
    for key in g.files:
        if key.startswith("grad_full."):
            k = key[len("grad_full."):]
            ref, got = g[key], grads[k].double().cpu().numpy()
        elif key.startswith("grad_sub."):
            k = key[len("grad_sub."):]
            ref, got = g[key], _sub(grads[k].contiguous())
        else:
            continue
        assert np.linalg.norm(got - ref) <= compat_run["l2_rtol"] * np.linalg.norm(ref), k
        # a ReLU whose pre-activation sits at rounding distance from 0 flips its mask and moves single
        # elements of a BN-bias gradient by a whole dy: bound the bulk, not the outliers
        assert np.quantile(np.abs(got - ref), 0.99) <= (0.15 if compat_run["gn_rtol"] > 0.1 else 0.1) * np.abs(ref).max(), k


def test_buffers_and_adamw(compat_run, golden_dir):
    g, model = compat_run["g"], compat_run["model"]
    sd = model.state_dict()
    keys = open(os.path.join(golden_dir, "state_dict_keys.txt")).read().split()
    assert list(sd.keys()) == keys
    bn_keys = [k for k in keys if k.endswith(("running_mean", "running_var"))]
    bs = np.array([sd[k].double().norm().item() for k in bn_keys])
    np.testing.assert_allclose(bs, g["post_buf_norm"], rtol=2e-4)
    nbt = np.array([int(sd[k]) for k in keys if k.endswith("num_batches_tracked")])
    np.testing.assert_array_equal(nbt, g["post_nbt"])
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    pn = np.array([sd[k].double().norm().item() for k in names])
    # step-1 Adam is g/(|g|+eps), sign-like: fp32 noise in near-zero gradients moves those parameters by
    # O(lr) (the fp32 CPU oracle shows the same 1e-3-level spread against the fp64 reference)
    np.testing.assert_allclose(pn, g["post_param_norm"], rtol=5e-3)
    # saved weights keep the reference's OIHW shape
    assert tuple(sd["derm_backbone.encoder.conv1.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["derm_backbone.encoder.layer1.0.conv2.weight"].shape) == (64, 64, 3, 3)


def test_extract_eval_T0(compat_run):
    g = compat_run["g"]
    fd, fc = compat_run["feats"]
    assert fd.dtype == torch.float32 and tuple(fd.shape) == g["extract_derm"].shape
    np.testing.assert_allclose(fd.double().cpu().numpy(), g["extract_derm"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fc.double().cpu().numpy(), g["extract_clinic"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
@pytest.mark.parametrize("case", ["b4_s64_f64", "b8_s64_style1_f64", "b3_s96_style2_f64"])
def test_extract_eval_16bit_against_the_reference_golden(golden_dir, case, dtname):
    """The 16-bit arithmetic modes against the REFERENCE's own numbers (not against this repository's f32 mode): the same
    sequence the goldens were generated with -- one train-mode forward (running statistics take their first momentum
    update), then `model.extract` in eval mode (src/models/simclr.py:399-413: frozen statistics, the fused conv + BatchNorm
    inference launches) -- in bf16 and fp16, features against the reference's fp64 features.  Measured relative L2 error
    (printed; reproducible on a build): bf16 3.0e-2 - 4.5e-2, fp16 5.0e-3 - 8.8e-3 over the three cases -- about 8x the
    figures of the 224 x 224 eval-mode test (tests/test_round5_gpu.py: 5.0e-3 / 6e-4), because the goldens' batches are 3 - 8
    images of 64 - 96 pixels (2 x 2 maps in layer 4) and the running statistics the eval pass normalises with come from one
    16-bit batch of that size.  Bounds: about twice the largest value seen."""
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    g = _load(golden_dir, case)
    batch, size, seed, style = [int(v) for v in g["meta"]]
    model = _build(seed, dt)
    model.train()
    derm, clinic = _batch(batch, size, seed)
    with torch.no_grad():
        model(derm, clinic, style)
    model.eval()
    with torch.no_grad():
        fd, fc = model.extract(derm[0], clinic[0])
    torch.cuda.synchronize()
    errs = []
    for got, key in ((fd, "extract_derm"), (fc, "extract_clinic")):
        ref = g[key]
        assert got.dtype == torch.float32 and tuple(got.shape) == ref.shape
        errs.append(float(np.linalg.norm(got.double().cpu().numpy() - ref) / np.linalg.norm(ref)))
    print(f"{case} {dtname}: extract features, relative L2 error against the reference's fp64 run: derm {errs[0]:.2e}, clinic {errs[1]:.2e}")
    assert max(errs) < (0.1 if dtname == "bf16" else 2e-2), errs


def test_fused_trainer_matches_golden_and_compat(golden_dir):
    """The fused step (no autograd, fused NT-Xent, fused AdamW) gives the reference's loss and post-step
    parameters."""
    from sm3hip.trainer import SM3Trainer
    g = _load(golden_dir, "b4_s64_f64")
    batch, size, seed, style = [int(v) for v in g["meta"]]
    model = _build(seed, torch.float32)
    derm, clinic = _batch(batch, size, seed)
    tr = SM3Trainer(model, lr=float(g["lr"]), weight_decay=5e-2, eps=1e-5, style=style)
    loss = tr.step(derm, clinic)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"])) < 1e-3
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    gv = dict(zip(names, tr._engine().store.grad_views()))
    gn = np.array([gv[k].double().norm().item() for k in names])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=5e-2, atol=1e-7)
    sd = model.state_dict()
    pn = np.array([sd[k].double().norm().item() for k in names])
    np.testing.assert_allclose(pn, g["post_param_norm"], rtol=5e-3)
    # optimizer state in torch.optim.AdamW's wire format
    osd = tr.optimizer_state_dict()
    assert len(osd["state"]) == len(names) and osd["param_groups"][0]["eps"] == 1e-5
    assert tuple(osd["state"][0]["exp_avg"].shape) == (64, 3, 7, 7)


def test_bf16_path_T1(golden_dir):
    from sm3hip.trainer import SM3Trainer
    g = _load(golden_dir, "b4_s64_f64")
    batch, size, seed, style = [int(v) for v in g["meta"]]
    derm, clinic = _batch(batch, size, seed)
    feats = {}
    for dt in (torch.float32, torch.bfloat16):
        model = _build(seed, dt)
        model.eval()
        with torch.no_grad():
            feats[dt] = model.extract(derm[0], clinic[0])[0].double()
    rel = (feats[torch.bfloat16] - feats[torch.float32]).norm() / feats[torch.float32].norm()
    assert float(rel) < 3e-2, float(rel)
    # The training step against the reference's fp64 goldens at B = 8 (style 1) and B = 32 (style 0).  NOT at the B = 4
    # golden: with 4 pairs the deepest BatchNorms normalise over 16 rows and the loss moves by O(1) with the placement of
    # 16-bit roundings -- measured there (golden 5.360): bf16 5.349 / 4.843 / 4.231 and fp16 6.346 / 6.341 / 6.160 for the
    # two-pass BatchNorm form / conv3 by linearity / + the downsample join (scratch/t1_variants.py), i.e. the *more*
    # precise fp16 type is off by 1.0 in every form.  From B = 8 on the 16-bit steps sit within 0.02-0.35 of fp64.
    for tag, bound16 in (("b8_s64_style1_f64", 0.5), ("b32_s64_f64", 0.5)):
        g = _load(golden_dir, tag)
        batch, size, seed, style = [int(v) for v in g["meta"]]
        derm, clinic = _batch(batch, size, seed)
        for dt, bound in ((torch.bfloat16, bound16), (torch.float16, 0.15)):
            model = _build(seed, dt)
            tr = SM3Trainer(model, lr=1e-3, style=style, init_scale=1024.0)
            loss = tr.step(derm, clinic)
            torch.cuda.synchronize()
            assert np.isfinite(float(loss)) and abs(float(loss) - float(g["loss"])) < bound, (tag, dt, float(loss), float(g["loss"]))
            assert bool(torch.isfinite(tr._engine().store.flat_g).all())
            del tr, model


def test_batch_permutation_invariance_224():
    """Size-independent property at the benchmark's image size: NT-Xent over BN-normalised projections is
    invariant to a permutation of the pairs in the batch (same permutation for all four views)."""
    from sm3hip.trainer import SM3Trainer
    B = 16
    g = torch.Generator(device="cpu").manual_seed(3407)
    derm = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)]
    clinic = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)]
    perm = torch.randperm(B, generator=g).cuda()
    losses, gnorms = [], []
    for permute in (False, True):
        model = _build(7, torch.float32)
        tr = SM3Trainer(model, lr=1e-4)
        d = [x[perm].contiguous() for x in derm] if permute else derm
        c = [x[perm].contiguous() for x in clinic] if permute else clinic
        losses.append(float(tr.step(d, c)))
        gnorms.append(float(tr._engine().store.flat_g.double().norm()))
    assert abs(losses[0] - losses[1]) < 2e-3, losses
    assert abs(gnorms[0] - gnorms[1]) < 5e-2 * gnorms[0], gnorms


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


def test_three_steps_carry_state_like_the_oracle():
    """Three consecutive fused steps with new inputs every step against the CPU oracle in fp64: BatchNorm running
    statistics, Adam moments, step count / bias correction carried across steps.

    The LOSS trajectory itself cannot be compared beyond the first update: at random init Adam's first steps move
    every weight by ~lr whatever the gradient scale, and the reference's own arithmetic run in fp32 instead of fp64
    lands 0.2 (step 1) and 0.7 (step 2) away in loss at B=4, lr=1e-4 (measured with the oracle, which reproduces
    the reference over these three steps to 5e-8).  So lr = 1e-7 keeps the weights where they are -- each step's
    loss is then comparable to 2e-3 -- and the state that accumulates is checked through the summed update
    p3 - p0 (direction and length per tensor) and the buffers."""
    from oracle import procedural, sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    seed, batch, size, lr = 11, 4, 64, 1e-7
    state = procedural.make_state_dict(seed=seed)
    P, B = O.split_state(state, torch.float64)
    p0 = {k: v.detach().clone() for k, v in P.items()}
    opt = {}
    model = _build(seed, torch.float32)
    tr = SM3Trainer(model, lr=lr, weight_decay=5e-2, eps=1e-5, style=0)
    for step in range(3):
        derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed + 100 * step)
        want, _ = O.train_step(P, B, [torch.from_numpy(a).double() for a in derm_np],
                               [torch.from_numpy(a).double() for a in clinic_np], 0, 0.1, opt, lr)
        got = tr.step([torch.from_numpy(a).cuda() for a in derm_np], [torch.from_numpy(a).cuda() for a in clinic_np])
        torch.cuda.synchronize()
        assert abs(float(got) - float(want)) < 2e-3, (step, float(got), float(want))
    sd = model.state_dict()
    for k in ("derm_backbone.encoder.bn1.running_mean", "clinic_backbone.encoder.layer4.2.bn3.running_var",
              "cross_proj.1.4.running_var"):
        np.testing.assert_allclose(sd[k].double().cpu().numpy(), B[k].numpy(), rtol=2e-3, atol=1e-5)
    assert int(sd["derm_backbone.encoder.bn1.num_batches_tracked"]) == 6  # two views x three steps
    cos, ratio = [], []
    for k, v0 in p0.items():
        d_or, d_hip = (P[k].detach() - v0).reshape(-1), (sd[k].double().cpu() - v0).reshape(-1)
        if float(d_or.norm()) < 1e-12:
            continue
        cos.append(float(d_or @ d_hip / (d_or.norm() * d_hip.norm())))
        ratio.append(float(d_hip.norm() / d_or.norm()))
    # fp32 master weights hold a 3e-7 update of an O(1e-2) weight to ~1e-2 relative: direction is the check
    assert min(cos) > 0.85 and float(np.mean(cos)) > 0.95, (min(cos), float(np.mean(cos)))
    assert 0.9 < float(np.median(ratio)) < 1.1, float(np.median(ratio))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_training_overfits_a_fixed_batch(dt):
    """Optimisation sanity beyond single-step parity.  Trajectories are chaotic at random init (see above) but the
    end state is not: stepping on ONE fixed batch of 16 learnable pairs (tools/backbone_train.py `latent` data) at
    lr = 3e-4 the reference arithmetic (CPU oracle, fp32) takes the 4-term NT-Xent loss from 11.6 to 0.09 in 15
    steps and 0.02 in 30; the HIP step must get there too, in both arithmetic modes."""
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("sm3_backbone_train", os.path.join(tools, "backbone_train.py"))
    bt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bt)
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(5)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.sm3_dtype = dt
    model.to("cuda:0")
    tr = SM3Trainer(model, lr=3e-4, weight_decay=5e-2, eps=1e-5, style=0)
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    derm, clinic = bt.synthetic_batch(16, (64, 64), torch.device("cuda:0"), gen, "latent")
    losses = [tr.step(derm, clinic) for _ in range(30)]
    torch.cuda.synchronize()
    first, last = float(losses[0]), float(sum(losses[-3:]) / 3)
    assert first > 5.0 and np.isfinite(last) and last < 0.3, (first, last, [round(float(v), 3) for v in losses])


def test_both_views_as_one_batch_equal_per_view_passes_and_the_oracle():
    """B = 32 at 64x64 is the smallest configuration in which every feature map of a view is a multiple of 128 rows,
    so the engine sends both views of a branch through the encoder as ONE batch of 2B images (per-view BatchNorm
    statistics).  Forward arithmetic is tile for tile the same as in two per-view passes -> identical loss and
    running statistics; weight gradients differ only by the summation order of the longer pixel axis; and the
    whole step matches the CPU oracle (fp64)."""
    from oracle import procedural, sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    seed, batch, size = 21, 32, 64
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    derm = [torch.from_numpy(a).cuda() for a in derm_np]
    clinic = [torch.from_numpy(a).cuda() for a in clinic_np]
    runs = {}
    for pair in (True, False):
        model = _build(seed, torch.float32)
        tr = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
        eng = tr._engine()
        eng.pair_views = pair
        assert eng.pair_ok(batch, size, size)
        loss = tr.step(derm, clinic)
        torch.cuda.synchronize()
        grads = [g.clone() for g in eng.store.grad_views()]
        sd = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
        runs[pair] = (float(loss), grads, sd)
    for k in runs[True][2]:
        assert torch.equal(runs[True][2][k], runs[False][2][k]), k
    assert runs[True][0] == runs[False][0]  # same forward bits + a fixed-order loss sum (no float atomics): equal
    for a, b in zip(runs[True][1], runs[False][1]):
        assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-9
    state = procedural.make_state_dict(seed=seed)
    P, B = O.split_state(state, torch.float64)
    want, _ = O.train_step(P, B, [torch.from_numpy(a).double() for a in derm_np],
                           [torch.from_numpy(a).double() for a in clinic_np], 0, 0.1)
    assert abs(runs[True][0] - float(want)) < 1e-3
    names = open(os.path.join(os.path.dirname(__file__), "golden", "param_names.txt")).read().split()
    gn = np.array([float(g.double().norm()) for g in runs[True][1]])
    on = np.array([float(P[k].grad.norm()) for k in names])
    np.testing.assert_allclose(gn, on, rtol=3e-2, atol=1e-7)
    assert int(runs[True][2]["derm_backbone.encoder.bn1.num_batches_tracked"]) == 2
    # the reference's literal call contract (model(...) -> CrossEntropyLoss -> backward) takes the same route
    model = _build(seed, torch.float32)
    model.train()
    outs = model(derm, clinic, 0)
    crit = torch.nn.CrossEntropyLoss()
    loss = crit(*outs[0]) + crit(*outs[1]) + 0.5 * crit(*outs[2][0]) + 0.5 * crit(*outs[2][1])
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - runs[True][0]) < 1e-4
    cn = np.array([float(p_.grad.double().norm()) for _, p_ in model.named_parameters()])
    np.testing.assert_allclose(cn, gn, rtol=2e-3, atol=1e-7)
