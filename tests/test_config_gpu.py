"""Round-2 parity cases (GPU) on the configurations BASELINE.json names, through the C ABI:

  * config 2 at its own size -- B = 256 pairs, 224x224, bf16, both views of a branch in one batch (512 images per
    launch, the largest tensors of the path): identical forward / running statistics to per-view passes, loss against
    the exact-f32 mode on the same inputs (direct stem there too since round 3: no im2col tensor), finite gradients;
  * T2: bf16 vs exact-f32 at random init and from a trained state, with PyTorch's own bf16 autocast of
    the same network (CPU oracle) as the yardstick -- loss, logits, gradient direction;
  * the B = 32 golden generated from the reference itself (oracle/gen_golden.py b32): BatchNorm1d over 32 / 64
    rows instead of 4 / 8, so the gradient bounds are the reference's own fp32-vs-fp64 spread at THIS size;
  * config 5's image size (448x448) at B = 32 in bf16;
  * f-3: checkpoint in the reference's wire format -> resume -> same third step.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _build(seed, dtype, state=None):
    from oracle import procedural
    from src.models.simclr import SimCLRSkinV32
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    if state is None:
        state = {k: torch.from_numpy(v) for k, v in procedural.make_state_dict(seed=seed).items()}
    model.load_state_dict(state, strict=True)
    model.sm3_dtype = dtype
    return model.to(DEV)


def _batch(batch, size, seed):
    from oracle import procedural
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    return ([torch.from_numpy(a).to(DEV) for a in derm_np], [torch.from_numpy(a).to(DEV) for a in clinic_np])


def _latent_batch(batch, size, seed):
    """tools/backbone_train.py's `latent` synthetic pairs: two views of a derm / clinic image share a latent."""
    import importlib.util
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")
    spec = importlib.util.spec_from_file_location("sm3_backbone_train", os.path.join(tools, "backbone_train.py"))
    bt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bt)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    return bt.synthetic_batch(batch, (size, size), torch.device(DEV), gen, "latent")


def _compat_step(model, derm, clinic, style=0):
    """The reference's literal loop body (tools/backbone_train.py:98-125) up to backward()."""
    crit = torch.nn.CrossEntropyLoss()
    outs = model(derm, clinic, style)
    w = 0.25 if style == 2 else 0.5
    loss = crit(*outs[0]) + crit(*outs[1]) + sum(w * crit(*o) for o in outs[2])
    model.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    logits = [outs[0][0].detach(), outs[1][0].detach()] + [o[0].detach() for o in outs[2]]
    return float(loss.detach()), logits, {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def test_config2_b256_224_bf16():
    """BASELINE.json configs[1] itself: what bench.py times."""
    from sm3hip.trainer import SM3Trainer
    B, S = 256, 224
    g = torch.Generator(device=DEV).manual_seed(3407)
    derm = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    clinic = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    torch.manual_seed(3407)
    from src.models.simclr import SimCLRSkinV32
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    runs = {}
    for key, dt, pair in (("bf16_pair", torch.bfloat16, True), ("bf16_views", torch.bfloat16, False),
                          ("f32", torch.float32, True)):
        model = _build(0, dt, init)
        tr = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
        eng = tr._engine()
        eng.pair_views = pair
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        gflat = eng.store.flat_g
        assert bool(torch.isfinite(gflat).all()), key
        stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
        runs[key] = (loss, float(gflat.double().norm()), stats)
        del tr, eng, model, gflat
        torch.cuda.empty_cache()
    # both views as ONE batch of 512 images == two passes of 256: forward arithmetic is tile for tile the same -> running
    # statistics bit-identical (asserted first: they pin the forward); the loss is a fixed-order fp64 sum of per-row terms
    # (round 5: no float atomics), so it is EQUAL too; weight gradients differ by the summation order of the pixel axis
    for k, v in runs["bf16_pair"][2].items():
        assert torch.equal(v, runs["bf16_views"][2][k]), k
    assert runs["bf16_pair"][0] == runs["bf16_views"][0], (runs["bf16_pair"][0], runs["bf16_views"][0])
    assert abs(runs["bf16_pair"][1] - runs["bf16_views"][1]) < 2e-3 * runs["bf16_views"][1]
    assert int(runs["bf16_pair"][2]["derm_backbone.encoder.bn1.num_batches_tracked"]) == 2
    # against the exact-f32 MFMA mode on the same inputs and weights.  Random init + N(0,1) noise images is the
    # worst-conditioned state there is (the reference's own bf16-autocast run is 0.2-0.85 off at B=16);
    # with BatchNorm1d over 256/512 rows the bf16 step lands within about 0.1 of it: -0.08 .. +0.10 over input / weight seeds
    # (round 3, general and lean epilogues alike; fp16: -0.026), so the bound is 0.25
    assert abs(runs["bf16_pair"][0] - runs["f32"][0]) < 0.25, (runs["bf16_pair"][0], runs["f32"][0])
    assert abs(runs["bf16_pair"][1] - runs["f32"][1]) < 0.15 * runs["f32"][1], (runs["bf16_pair"][1], runs["f32"][1])
    for k, v in runs["f32"][2].items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            d = (runs["bf16_pair"][2][k].double() - v.double()).norm() / (v.double().norm() + 1e-12)
            assert float(d) < 2e-2, (k, float(d))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_b512_both_views_one_batch(dt):
    """B = 512 pairs per GPU: 1 024 images per launch, activation tensors up to 1.6 GB.  Possible in the one-batch mode only
    since the stem reads the images directly (the 2B-image im2col matrix would pass the 3 GB buffer-offset limit); equal
    to the per-view passes (running statistics and the loss bit for bit)."""
    from sm3hip.trainer import SM3Trainer
    B, S = 512, 224
    g = torch.Generator(device=DEV).manual_seed(11)
    derm = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    clinic = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    torch.manual_seed(11)
    from src.models.simclr import SimCLRSkinV32
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    runs = {}
    for pair in (True, False):
        model = _build(0, dt, init)
        tr = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
        eng = tr._engine()
        assert eng.pair_ok(B, S, S)
        eng.pair_views = pair
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eng.store.flat_g).all())
        runs[pair] = (loss, float(eng.store.flat_g.double().norm()),
                      {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
        del tr, eng, model
        torch.cuda.empty_cache()
    # the forward is pinned bit for bit by the running statistics; the loss (four NT-Xent terms, each a fixed-order sum of
    # its per-row terms) is then equal as well
    for k, v in runs[True][2].items():
        assert torch.equal(v, runs[False][2][k]), k
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert abs(runs[True][1] - runs[False][1]) < 2e-3 * runs[False][1]


def _flat(grads, names):
    return torch.cat([grads[k].detach().double().cpu().flatten() for k in names])


@pytest.mark.parametrize("steps", [0, 10], ids=["random_init", "after_10_steps"])
def test_T2_bf16_step_against_f32_with_torch_autocast_as_yardstick(steps):
    """SURVEY.md 8c T2.  One step of the reference's literal loop in bf16 and in exact f32 from the same state on the
    same batch of 32 learnable (`latent`) pairs -- at random init and after 10 fused f32 steps (loss 14 -> ~0.5).

    What bf16 can deliver here is not a free parameter: the 1/0.1 temperature behind BatchNorm1d(affine=False) +
    L2-normalise turns a 3 % feature error into logit errors of ~0.5-1 (range +-10), and the gradient follows the
    softmax weights.  PyTorch's OWN bf16 autocast of this network (the CPU oracle under torch.autocast -- what the
    reference would compute with --amp in bf16) lands at: loss off by 0.49, logits rms 0.96, gradient cosine 0.15
    against fp32 at B = 32 on the procedural batch (measured; the reference's real AMP mode, fp16, gives 0.022 / 0.26 /
    0.61); on this test's states: logits rms 0.93 / cosine 0.12 at random init and 0.68 / 0.31 after 10 steps, where
    the HIP bf16 step measures 0.87 / 0.17 and 0.57 / 0.35.  So the
    yardstick is that run, on the same state: the HIP bf16 step must be no further from f32 than torch's bf16
    autocast is, and the HIP f32 step must agree with the fp32 oracle outright."""
    from oracle import sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(5)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.sm3_dtype = torch.float32
    model.to(DEV)
    derm, clinic = _latent_batch(32, 64, 7)
    if steps:
        tr = SM3Trainer(model, lr=3e-4, weight_decay=5e-2, eps=1e-5, style=0)
        for _ in range(steps):
            tr.step(derm, clinic)
        del tr
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    names = [k for k in state if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    hip = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _build(0, dt, state)
        m.train()
        loss, logits, grads = _compat_step(m, derm, clinic)
        hip[dt] = (loss, torch.cat([l.double().cpu().flatten() for l in logits]), _flat(grads, names))
        del m
    # CPU oracle: fp32, and the same code under torch's bf16 autocast
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = {}
    state_np = {k: v.cpu().numpy() for k, v in state.items()}
    dc, cc = [t.cpu() for t in derm], [t.cpu() for t in clinic]
    for key, amp in (("f32", None), ("bf16", torch.bfloat16)):
        P, Bf = O.split_state(state_np, torch.float32)
        with torch.autocast("cpu", dtype=amp, enabled=amp is not None):
            outs = O.sm3_v32_forward(P, Bf, dc, cc, 0, 0.1, True, None)
            loss = O.sm3_loss(outs, 0)
        loss.backward()
        lg = [outs[0][0], outs[1][0]] + [o[0] for o in outs[2]]
        cpu[key] = (float(loss), torch.cat([l.detach().double().flatten() for l in lg]),
                    torch.cat([P[k].grad.double().flatten() for k in names]))

    def dist(a, b):
        cos = float(a[2] @ b[2] / (a[2].norm() * b[2].norm()))
        return abs(a[0] - b[0]), float((a[1] - b[1]).pow(2).mean().sqrt()), cos

    f32_vs_oracle = dist(hip[torch.float32], cpu["f32"])
    hip_bf16 = dist(hip[torch.bfloat16], hip[torch.float32])
    torch_bf16 = dist(cpu["bf16"], cpu["f32"])
    print(f"T2[{steps}]: (|dloss|, logit rms, grad cosine)  HIP f32 vs oracle f32 {f32_vs_oracle};  "
          f"HIP bf16 vs HIP f32 {hip_bf16};  torch bf16 autocast vs fp32 {torch_bf16}")
    assert f32_vs_oracle[0] < 1e-3 and f32_vs_oracle[1] < 2e-3 and f32_vs_oracle[2] > 0.999, f32_vs_oracle
    # the loss is one scalar: either run can land close to f32 by luck (torch's did at random init: 0.007 against a
    # logit rms of 0.93), so it only gets an absolute bound; logits and gradient direction carry the comparison
    assert hip_bf16[0] <= max(1.5 * torch_bf16[0], 0.3), (hip_bf16, torch_bf16)
    assert hip_bf16[1] <= 1.25 * torch_bf16[1], (hip_bf16, torch_bf16)
    # the yardstick's own gradient cosine moves by ~0.14 between hosts / runs on the after-10-steps state (measured: 0.313 and
    # 0.451 for torch's CPU bf16 autocast, with the HIP bf16 step at 0.350 and 0.371 on the same two runs), hence the margin
    assert hip_bf16[2] >= torch_bf16[2] - 0.15, (hip_bf16, torch_bf16)
    assert hip_bf16[0] < 0.6 and hip_bf16[2] > 0.1, hip_bf16  # absolute sanity, whatever the yardstick says


def test_b32_golden_from_the_reference(golden_dir):
    """The exact-f32 MFMA path against the reference's own fp64 run at B = 32 (oracle/gen_golden.py b32)."""
    g = np.load(os.path.join(golden_dir, "sm3_v32_b32_s64_f64.npz"))
    batch, size, seed, style = [int(v) for v in g["meta"]]
    model = _build(seed, torch.float32)
    model.train()
    derm, clinic = _batch(batch, size, seed)
    loss, logits, grads = _compat_step(model, derm, clinic, style)
    assert abs(loss - float(g["loss"])) < 3e-4, (loss, float(g["loss"]))
    for got, key in zip(logits, ("derm_logits", "clinic_logits", "cross_logits_0", "cross_logits_1")):
        np.testing.assert_allclose(got.cpu().double().numpy(), g[key], atol=2e-3, rtol=0)
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    gn = np.array([grads[k].double().norm().item() for k in names])
    # Bounds: the reference's own fp32 run against its fp64 run at this size (printed by gen_golden.py b32) is
    # 1.3e-5 in loss, 4e-4 in logits, 4.1e-3 in the worst gradient norm and 1.2e-2 .. 1.7e-2 relative L2 in the
    # convolution weight gradients -- B = 32 removes the BatchNorm1d-over-4-rows pathology but not fp32 itself, so
    # 1e-3 on gradients is below what the reference's arithmetic delivers; the bounds here are 2x that floor.
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=B32_GN_RTOL, atol=1e-7)
    for key in g.files:
        if key.startswith("grad_full."):
            k = key[len("grad_full."):]
            ref, got = g[key], grads[k].double().cpu().numpy()
        elif key.startswith("grad_sub."):
            k = key[len("grad_sub."):]
            flat = grads[k].contiguous().reshape(-1)
            step = max(1, flat.numel() // 256)
            ref, got = g[key], flat[::step][:256].double().cpu().numpy()
        else:
            continue
        assert np.linalg.norm(got - ref) <= B32_L2_RTOL * np.linalg.norm(ref), (k, np.linalg.norm(got - ref) / np.linalg.norm(ref))
    model.eval()
    with torch.no_grad():
        fd, fc = model.extract(derm[0], clinic[0])
    np.testing.assert_allclose(fd[:8].double().cpu().numpy(), g["extract_derm"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fc[:8].double().cpu().numpy(), g["extract_clinic"], rtol=1e-3, atol=1e-3)


B32_GN_RTOL = 1e-2
B32_L2_RTOL = 3.5e-2


def test_config5_image_size_448_b32_16bit_modes():
    """BASELINE.json configs[4] -- 448x448, fp16 + loss scaling -- at B = 32: both views in one batch (14x14 maps x 32
    images = 49 row tiles per view); the fp16 mode (dynamic loss scaling on the device, as the config is written) and the
    bf16 mode against the exact-f32 mode, finite gradients, the fp16 step taken (no overflow at init_scale 65536)."""
    from sm3hip.trainer import SM3Trainer
    B, S = 32, 448
    derm, clinic = _latent_batch(B, S, 11)
    torch.manual_seed(11)
    from src.models.simclr import SimCLRSkinV32
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    out = {}
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        model = _build(0, dt, init)
        tr = SM3Trainer(model, lr=1e-6)
        eng = tr._engine()
        assert eng.pair_ok(B, S, S)
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eng.store.flat_g).all())
        gn = float(eng.store.flat_g.double().norm())
        if dt == torch.float16:
            assert tr.steps_taken() == 1 and float(tr._scaler["scale"]) == 65536.0
            gn /= 65536.0
        out[dt] = (loss, gn)
        del tr, eng, model
        torch.cuda.empty_cache()
    # random init at B = 32: the loss is only comparable to ~1 in bf16 (see T2 above: torch's own bf16 autocast of this
    # network is 0.5 off at this batch size); the gradient norm is the tighter signal; fp16 sits closer on both
    assert abs(out[torch.bfloat16][0] - out[torch.float32][0]) < 1.0, out
    assert abs(out[torch.bfloat16][1] - out[torch.float32][1]) < 0.1 * out[torch.float32][1], out
    assert abs(out[torch.float16][0] - out[torch.float32][0]) < 0.5, out
    assert abs(out[torch.float16][1] - out[torch.float32][1]) < 0.05 * out[torch.float32][1], out


def test_checkpoint_resume_in_the_reference_wire_format(tmp_path):
    """f-3 (tools/backbone_train.py:575-592, src/utils/misc.py:462-494): two fused steps -> torch.save of
    {"epoch", "state_dict", "optimizer", "scaler"} -> a FRESH model + load_state_dict + load_optimizer_state_dict ->
    the restored state is bit-identical to the saved one and the third step equals the uninterrupted third step
    (to the order noise of the float-atomic weight-gradient sums; parameters move by at most one Adam update)."""
    from sm3hip.trainer import SM3Trainer
    lr, seed, batch, size = 1e-4, 31, 8, 64
    batches = [_batch(batch, size, seed + i) for i in range(3)]
    model = _build(seed, torch.float32)
    tr = SM3Trainer(model, lr=lr, weight_decay=5e-2, eps=1e-5, style=0)
    for i in range(2):
        tr.step(*batches[i])
    torch.cuda.synchronize()
    path = str(tmp_path / "checkpoint.pth.tar")
    torch.save({"epoch": 1, "state_dict": model.state_dict(), "optimizer": tr.optimizer_state_dict(), "scaler": {}}, path)
    saved_flat = tr._engine().store.flat_p.clone()
    saved_m, saved_v = tr.m.clone(), tr.v.clone()
    loss3 = float(tr.step(*batches[2]))
    torch.cuda.synchronize()
    p3 = tr._engine().store.flat_p.clone()
    bufs3 = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}

    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "optimizer", "scaler"}
    keys = open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.txt")).read().split()
    assert list(ck["state_dict"].keys()) == keys
    assert tuple(ck["state_dict"]["derm_backbone.encoder.layer1.0.conv2.weight"].shape) == (64, 64, 3, 3)
    from src.models.simclr import SimCLRSkinV32
    fresh = SimCLRSkinV32("resnet50", None, 128, 0.1)
    fresh.sm3_dtype = torch.float32
    missing = fresh.load_state_dict(ck["state_dict"], strict=False)  # misc.py:476-487 loads with strict=False
    assert not missing.missing_keys and not missing.unexpected_keys
    fresh.to(DEV)
    tr2 = SM3Trainer(fresh, lr=123.0, weight_decay=0.0, eps=1.0, style=0)  # every hyper-parameter comes from the checkpoint
    tr2.load_optimizer_state_dict(ck["optimizer"])
    assert (tr2.lr, tr2.wd, tr2.eps, tr2.step_count) == (lr, 5e-2, 1e-5, 2)
    assert torch.equal(tr2._engine().store.flat_p, saved_flat)
    assert torch.equal(tr2.m, saved_m) and torch.equal(tr2.v, saved_v)
    loss3b = float(tr2.step(*batches[2]))
    torch.cuda.synchronize()
    assert abs(loss3 - loss3b) < 1e-5, (loss3, loss3b)
    d = (tr2._engine().store.flat_p - p3).abs()
    assert float(d.max()) <= 2.0 * lr and float(d.mean()) < 2e-2 * lr, (float(d.max()), float(d.mean()))
    for k, v in fresh.state_dict().items():
        if k in bufs3:
            assert torch.allclose(v, bufs3[k], rtol=1e-5, atol=1e-6), k
    assert int(fresh.state_dict()["derm_backbone.encoder.bn1.num_batches_tracked"]) == 6


def test_torch_adamw_checkpoint_resumes_in_the_fused_engine():
    """A checkpoint written by the compat path (the reference's loop with torch.optim.AdamW, whose state_dict() is what
    tools/backbone_train.py:582 saves) resumes in the fused trainer: the next fused step equals the next torch step."""
    from sm3hip.trainer import SM3Trainer
    lr, seed, batch, size = 1e-4, 33, 8, 64
    batches = [_batch(batch, size, seed + i) for i in range(3)]
    model = _build(seed, torch.float32)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=5e-2, eps=1e-5)
    crit = torch.nn.CrossEntropyLoss()

    def torch_step(derm, clinic):
        outs = model(derm, clinic, 0)
        loss = crit(*outs[0]) + crit(*outs[1]) + 0.5 * crit(*outs[2][0]) + 0.5 * crit(*outs[2][1])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss.detach())

    for i in range(2):
        torch_step(*batches[i])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    import copy
    osd = copy.deepcopy(opt.state_dict())  # state_dict() hands out the live tensors
    loss3 = torch_step(*batches[2])
    torch.cuda.synchronize()
    p3 = {k: v.detach().clone() for k, v in model.named_parameters()}

    fresh = _build(0, torch.float32, sd)
    tr = SM3Trainer(fresh, lr=0.0)
    tr.load_optimizer_state_dict(osd)
    assert tr.step_count == 2 and tr.lr == lr
    loss3b = float(tr.step(*batches[2]))
    torch.cuda.synchronize()
    assert abs(loss3 - loss3b) < 1e-5, (loss3, loss3b)
    dmax = max(float((p.detach() - p3[k]).abs().max()) for k, p in fresh.named_parameters())
    assert dmax <= 2.0 * lr, dmax
    num = sum(float((p.detach() - p3[k]).abs().sum()) for k, p in fresh.named_parameters())
    assert num / sum(p.numel() for p in fresh.parameters()) < 2e-2 * lr


def test_eval_mode_backward_through_an_encoder():
    """module.eval() with trainable parameters (frozen BatchNorm statistics inside an autograd graph): gradients
    against the CPU oracle's eval-mode forward."""
    from oracle import procedural, sm3_oracle as O
    import resnet
    state = procedural.make_state_dict(procedural.resnet50_spec(""), seed=17)
    x = torch.from_numpy(procedural.make_images(4, 64, 17, "derm0"))
    P, Bf = O.split_state(state, torch.float64)
    f_ref = O.resnet50_features(x.double(), P, Bf, "", False)
    (f_ref ** 2).sum().backward()
    m = resnet.resnet50(weights=None)
    m.fc = torch.nn.Identity()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = torch.float32
    m.to(DEV).eval()
    f = m(x.to(DEV))
    (f.double() ** 2).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(f.detach().cpu().double().numpy(), f_ref.detach().numpy(), rtol=1e-3, atol=1e-3)
    for k, p in m.named_parameters():
        ref = P[k].grad
        err = float((p.grad.double().cpu() - ref).norm()) / (float(ref.norm()) + 1e-12)
        assert err < 2e-3, (k, err)
    assert int(m.state_dict()["bn1.num_batches_tracked"]) == 0


def test_train_mode_encoder_gradients_with_torch_fp32_as_yardstick():
    """How tight can train-mode (batch-statistics) gradients be pinned?  Through 53 BatchNorms at random init the
    gradient of sum(f^2) is ill-conditioned for ANY fp32 arithmetic: PyTorch's own fp32 run of the oracle differs from
    its fp64 run by 1.6 % (median over the 161 tensors) / 2.1 % (worst) at B = 16, 64x64, while the forward agrees to
    3e-5.  The exact-f32 HIP path measures 0.9 % / 1.4 % -- fp64 statistics and f32 MFMA accumulation.  So the bound is
    the yardstick: never further from fp64 than torch's fp32 is, tensor by tensor in aggregate, and the forward tight.
    (The eval-mode twin of this test, frozen statistics, holds 2e-3 per tensor: test_eval_mode_backward_...)"""
    from oracle import procedural, sm3_oracle as O
    import resnet
    B, S = 16, 64
    state = procedural.make_state_dict(procedural.resnet50_spec(""), seed=23)
    x = torch.from_numpy(procedural.make_images(B, S, 23, "derm0"))
    grads = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        P, Bf = O.split_state(state, dt)
        f = O.resnet50_features(x.to(dt), P, Bf, "", True)
        (f ** 2).sum().backward()
        grads[name] = {k: v.grad.double() for k, v in P.items()}
        if name == "f64":
            f_ref = f.detach()
    m = resnet.resnet50(weights=None)
    m.fc = torch.nn.Identity()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = torch.float32
    m.to(DEV).train()
    f = m(x.to(DEV))
    (f.double() ** 2).sum().backward()
    torch.cuda.synchronize()
    assert float((f.detach().cpu().double() - f_ref).norm() / f_ref.norm()) < 2e-4
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    hip = sorted(rel(p.grad.double().cpu(), grads["f64"][k]) for k, p in m.named_parameters())
    tch = sorted(rel(grads["f32"][k], grads["f64"][k]) for k in grads["f64"])
    assert hip[len(hip) // 2] <= 1.25 * tch[len(tch) // 2] + 1e-3, (hip[len(hip) // 2], tch[len(tch) // 2])
    assert hip[-1] <= 1.5 * tch[-1] + 1e-3, (hip[-1], tch[-1])


def test_fp16_mode_with_dynamic_loss_scaling():
    """The reference's own AMP recipe (fp16 autocast + GradScaler, tools/backbone_train.py:27,98,125-127,480) as an
    arithmetic mode of the engine: fp16 storage + f16 MFMA, loss scaling with GradScaler's state on the device.

      * one step from random init: loss against the exact-f32 mode and against PyTorch's own fp16 autocast of the oracle
        (what the reference computes with --amp) -- fp16 keeps 11 bits where bf16 keeps 8, so it must land closer to
        f32 than the bf16 mode does;
      * an injected overflow: the step is skipped (parameters and Adam moments untouched, Adam's step count not
        advanced), the scale halves; the next clean step is taken;
      * growth: after `growth_interval` clean steps the scale doubles;
      * the compat path under torch.amp.GradScaler itself (the reference's literal loop) agrees with the fused step."""
    from oracle import sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    from sm3hip import ops
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(5)
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    derm, clinic = _latent_batch(32, 64, 7)
    names = [k for k in init if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    out = {}
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        m = _build(0, dt, init)
        tr = SM3Trainer(m, lr=1e-4, init_scale=1024.0, growth_interval=2)
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        eng = tr._engine()
        g = eng.store.flat_g.double().cpu()
        if dt == torch.float16:
            assert tr._scaler is not None and float(tr._scaler["scale"]) == 1024.0 and tr.steps_taken() == 1
            g = g / 1024.0  # flat_g holds the scaled gradient
        else:
            assert tr._scaler is None
        out[dt] = (loss, g, tr, m)
    cos = {dt: float(out[dt][1] @ out[torch.float32][1] / (out[dt][1].norm() * out[torch.float32][1].norm())) for dt in out}
    dl = {dt: abs(out[dt][0] - out[torch.float32][0]) for dt in out}
    print(f"fp16 mode: |dloss| bf16 {dl[torch.bfloat16]:.4f} f16 {dl[torch.float16]:.4f}; grad cosine vs f32: bf16 "
          f"{cos[torch.bfloat16]:.3f} f16 {cos[torch.float16]:.3f}")
    # measured: |dloss| 0.17 (bf16 0.29), gradient cosine 0.76 (bf16 0.13) at random init, B = 32
    assert dl[torch.float16] < 0.3 and cos[torch.float16] > cos[torch.bfloat16] + 0.1, (dl, cos)
    # PyTorch's fp16 autocast of the oracle on the same state (CPU): the HIP fp16 step is at least as close to f32
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    state_np = {k: v.cpu().numpy() for k, v in init.items()}
    dc, cc = [t.cpu() for t in derm], [t.cpu() for t in clinic]
    ref = {}
    for key, amp in (("f32", None), ("f16", torch.float16)):
        P, Bf = O.split_state(state_np, torch.float32)
        with torch.autocast("cpu", dtype=amp, enabled=amp is not None):
            loss = O.sm3_loss(O.sm3_v32_forward(P, Bf, dc, cc, 0, 0.1, True, None), 0)
        loss.backward()
        ref[key] = (float(loss), torch.cat([P[k].grad.double().flatten() for k in names]))
    cos_torch = float(ref["f16"][1] @ ref["f32"][1] / (ref["f16"][1].norm() * ref["f32"][1].norm()))
    print(f"           torch fp16 autocast vs fp32: |dloss| {abs(ref['f16'][0] - ref['f32'][0]):.4f}, grad cosine {cos_torch:.3f}")
    assert cos[torch.float16] >= cos_torch - 0.05, (cos, cos_torch)

    # ---- overflow: skip + back-off, then a clean step; growth after 2 clean steps ----
    _, _, tr, m = out[torch.float16]
    eng = tr._engine()
    p_before, m_before = eng.store.flat_p.clone(), tr.m.clone()
    big = [d * 3.0e4 for d in derm]  # activations beyond fp16's range -> inf in the forward, nan gradients
    tr.step(big, clinic)
    torch.cuda.synchronize()
    assert torch.equal(eng.store.flat_p, p_before) and torch.equal(tr.m, m_before)
    assert float(tr._scaler["scale"]) == 512.0 and tr.steps_taken() == 1 and int(tr._scaler["found_inf"]) == 0
    tr.step(derm, clinic)
    tr.step(derm, clinic)
    torch.cuda.synchronize()
    assert tr.steps_taken() == 3 and float(tr._scaler["scale"]) == 1024.0  # two clean steps: growth
    assert not torch.equal(eng.store.flat_p, p_before) and bool(torch.isfinite(eng.store.flat_p).all())
    sd = tr.scaler_state_dict()
    assert sd["scale"] == 1024.0 and sd["growth_interval"] == 2

    # ---- compat path under torch's GradScaler (the reference's literal loop) ----
    mc = _build(0, torch.float16, init)
    mc.train()
    opt = torch.optim.AdamW(mc.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    crit = torch.nn.CrossEntropyLoss()
    outs = mc(derm, clinic, 0)
    loss = crit(*outs[0]) + crit(*outs[1]) + 0.5 * crit(*outs[2][0]) + 0.5 * crit(*outs[2][1])
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - out[torch.float16][0]) < 2e-3
    assert scaler.get_scale() == 1024.0


def test_global_negatives_mode_single_rank_equals_local_and_rect_kernel():
    """Opt-in north_star mode (all-gather of projection embeddings).  With one rank the gathered set IS the local set, so
    the rectangular path (normalise -> exact-f32 GEMM -> sm3_ntxent_rect -> two gradient GEMMs -> normalise backward) must
    reproduce the fused local NT-Xent step; and sm3_ntxent_rect alone against torch on a 3x wider candidate set."""
    from sm3hip import ops
    from sm3hip.trainer import SM3Trainer
    torch.manual_seed(3)
    from src.models.simclr import SimCLRSkinV32
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    derm, clinic = _latent_batch(16, 64, 5)
    out = {}
    for gn in (False, True):
        m = _build(0, torch.float32, init)
        tr = SM3Trainer(m, lr=1e-4, global_negatives=gn)
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        out[gn] = (loss, tr._engine().store.flat_g.double().clone())
    assert abs(out[True][0] - out[False][0]) < 1e-5, (out[True][0], out[False][0])
    rel = float((out[True][1] - out[False][1]).norm() / out[False][1].norm())
    assert rel < 1e-3, rel
    # the kernel alone: 24 local rows (B = 12) at offset 48 of 96 candidates
    g = torch.Generator().manual_seed(9)
    Rl, Rg, off, T, w = 24, 96, 48, 0.1, 0.5
    zg = torch.nn.functional.normalize(torch.randn(Rg, 128, generator=g), dim=1).double()
    S = (zg[off:off + Rl] @ zg.t()).requires_grad_(True)
    idx = torch.arange(Rl)
    mask = torch.zeros(Rl, Rg, dtype=torch.bool)
    mask[idx, off + idx] = True
    ref = w * (torch.logsumexp((S / T).masked_fill(mask, float("-inf")), 1) - S[idx, off + (idx + Rl // 2) % Rl] / T).mean()
    ref.backward()
    Sd = S.detach().float().to(DEV).contiguous()
    loss = torch.zeros(1, device=DEV)
    ops.ntxent_rect(Sd, off, T, w, loss)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 1e-5
    assert float((Sd.cpu().double() - S.grad).abs().max()) < 1e-6


def test_momentum_target_extension():
    """north_star's momentum-updated target encoders (an extension: the reference has none).  With momentum 0 and the
    target initialised to the online weights the keys equal the queries, so the symmetrised query/key loss equals the
    plain loss; the EMA buffer follows the online weights by sm3_ema_update; gradients flow to the query rows only."""
    from sm3hip.trainer import SM3Trainer
    from sm3hip import ops
    torch.manual_seed(4)
    from src.models.simclr import SimCLRSkinV32
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    derm, clinic = _latent_batch(16, 64, 6)
    m0 = _build(0, torch.float32, init)
    base = SM3Trainer(m0, lr=1e-4)
    l0 = float(base.step(derm, clinic))
    m1 = _build(0, torch.float32, init)
    tr = SM3Trainer(m1, lr=1e-4, target_momentum=0.9)
    l1 = float(tr.step(derm, clinic))
    torch.cuda.synchronize()
    assert abs(l0 - l1) < 1e-4, (l0, l1)  # first step: target == online
    eng = tr._engine()
    # the target moved 10 % of the way to the updated online weights
    p_on = eng.store.flat_p
    p_init = base._engine().store.flat_p  # same init, same first update -> equals p_on; rebuild the initial point instead
    init_flat = SM3Trainer(_build(0, torch.float32, init), lr=0.0)._engine()
    init_flat.prepare(torch.device(DEV))
    want = 0.9 * init_flat.store.flat_p + 0.1 * p_on
    assert torch.allclose(tr.flat_target, want, rtol=1e-6, atol=1e-8)
    # BatchNorm buffers were updated once per view by the ONLINE pass only
    assert int(m1.state_dict()["derm_backbone.encoder.bn1.num_batches_tracked"]) == 2
    # second step: keys now differ from queries; loss finite, gradient only half as strong through the query rows
    l2 = float(tr.step(derm, clinic))
    torch.cuda.synchronize()
    assert np.isfinite(l2) and bool(torch.isfinite(eng.store.flat_g).all())
    # ema kernel alone
    a, b = torch.randn(1003, device=DEV), torch.randn(1003, device=DEV)
    a = torch.cat([a, torch.zeros(1, device=DEV)])[:1003].contiguous()
    t = torch.zeros(1008, device=DEV)[:1003]
    t.copy_(a)
    ops.ema_update(t, b, 0.75)
    torch.cuda.synchronize()
    assert torch.allclose(t, 0.75 * a + 0.25 * b, rtol=1e-6, atol=1e-7)


def test_metadata_mlp_extension_against_the_oracle():
    """north_star's metadata-MLP branch (an extension: no counterpart in the reference).  SimCLRSkinV32(metadata_dim=20)
    owns meta_proj; SM3Trainer.step(..., metadata=) adds 1/2 NT-Xent(cat(cross_proj[0](derm_f0), meta_proj(meta))) +
    1/2 NT-Xent(cat(cross_proj[1](clinic_f0), meta_proj(meta))).  Loss and gradients against the same definition written
    with the CPU oracle's pieces in fp64; the default model's state_dict is untouched by the option."""
    from oracle import procedural, sm3_oracle as O
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32, METADATA_PAD
    B, size, seed, T = 16, 64, 41, 0.1
    model = SimCLRSkinV32("resnet50", None, 128, T, metadata_dim=20)
    keys = open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.txt")).read().split()
    sd0 = model.state_dict()
    assert [k for k in sd0 if not k.startswith("meta_proj.")] == keys and len(sd0) == len(keys) + 16
    spec = [(k, tuple(v.shape)) for k, v in sd0.items()]
    state = procedural.make_state_dict(spec, seed=seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model.sm3_dtype = torch.float32
    model.to(DEV)
    derm_np, clinic_np = procedural.make_pair_batch(B, size, seed)
    g = torch.Generator().manual_seed(seed)
    meta = torch.randn(B, 20, generator=g)
    tr = SM3Trainer(model, lr=1e-4)
    loss = float(tr.step([torch.from_numpy(a).to(DEV) for a in derm_np], [torch.from_numpy(a).to(DEV) for a in clinic_np],
                         metadata=meta.to(DEV)))
    torch.cuda.synchronize()
    eng = tr._engine()
    grads = dict(zip(eng.store.names, eng.store.grad_views()))
    # oracle (fp64)
    P, Bf = O.split_state(state, torch.float64)
    derm = [torch.from_numpy(a).double() for a in derm_np]
    clinic = [torch.from_numpy(a).double() for a in clinic_np]
    zs = O.sm3_v32_projections(P, Bf, derm, clinic, 0, True)
    x = torch.zeros(B, METADATA_PAD, dtype=torch.float64)
    x[:, :20] = meta.double()
    zm = O.projector(x, P, Bf, "meta_proj.", True)
    nt = lambda z: O.ntxent_loss_closed_form(z, T)
    ref = (nt(zs[0]) + nt(zs[1]) + 0.5 * nt(zs[2]) + 0.5 * nt(zs[3])
           + 0.5 * nt(torch.cat([zs[2][:B], zm], 0)) + 0.5 * nt(torch.cat([zs[2][B:], zm], 0)))
    ref.backward()
    assert abs(loss - float(ref)) < 1e-3, (loss, float(ref))
    for k in ("meta_proj.0.weight", "meta_proj.3.weight", "meta_proj.6.weight", "meta_proj.1.weight", "meta_proj.4.bias",
              "cross_proj.0.6.weight", "cross_proj.1.0.weight", "derm_backbone.encoder.layer4.2.conv3.weight"):
        a, b = grads[k].double().cpu(), P[k].grad
        if k == "meta_proj.0.weight":  # the padded input columns receive no gradient
            assert float(a[:, 20:].abs().max()) == 0.0
        rel = float((a - b).norm() / b.norm())
        assert rel < 6e-2, (k, rel)
    assert int(model.state_dict()["meta_proj.1.num_batches_tracked"]) == 1
