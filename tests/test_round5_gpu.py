"""Round 5: an anchor for the BENCHMARKED arithmetic (bf16) that can fail.

The whole-step comparisons of bf16 against f32 (tests/test_round3_gpu.py, test_round4_gpu.py) go through the 1 / 0.1
temperature behind BatchNorm1d + L2-normalise: a 3 % feature error becomes an O(1) logit error and the gradient's direction
is lost at ANY state (whole-gradient cosine 0.1 - 0.4, torch's own bf16 autocast included), so their floors (0.15) pass
almost anything.  What those floors cannot see is a tap, halo or stride addressing bug confined to one stage of the encoder.
Two comparisons of a 16-bit ResNet-50 encoder against the exact-f32 encoder, tensor by tensor, at the network's real map
sizes (224 x 224: the halo-resident 3x3 kernels, the pointwise 1x1 variants, the strided data gradients, the direct stem --
what bench.py runs), same weights, same images, same upstream gradient, NO loss in the loop:

  * frozen BatchNorm statistics (eval mode with trainable parameters): well conditioned -- bf16 agrees with f32 to a cosine
    of 0.9998 or better in EVERY parameter tensor, fp16 to 0.99999.  This is the anchor that can fail.
  * batch statistics (train mode): what was measured is that taking the loss out does NOT restore bf16's direction (cosine
    0.14 over the whole gradient, about 0 for some BatchNorm tensors) while fp16 keeps 0.67 - 0.81 per stage: the
    amplification sits in the 53 train-mode BatchNorms themselves.  fp16 (same kernel templates) carries per-stage floors,
    bf16 the magnitudes.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _encoder_grads(dtype, x, dfeat, views):
    from src.models import resnet
    from sm3hip.engine import SM3Engine
    torch.manual_seed(5)
    m = resnet.resnet50()
    m.fc = torch.nn.Identity()
    m.to(DEV).train()
    eng = SM3Engine(m, dtype=dtype, kind="encoder")
    eng.prepare(torch.device(DEV))
    eng.refresh_weights()
    plan = eng.branches["main"][0]
    N = x.shape[0]
    feats = torch.empty(N, 2048, device=DEV)
    ctx = []
    eng.encoder_forward(plan, x, True, feats, None, ctx, views=views)
    eng.store.flat_g.zero_()
    eng.encoder_backward(ctx[0], dfeat.to(eng.tdt))
    torch.cuda.synchronize()
    st = eng.store
    grads = {n: st._view(st.flat_g, n).double().clone() for n in st.names}
    return feats.double().clone(), grads


def _stage(name):
    if name.startswith("layer"):
        return name.split(".")[0]
    return "stem"


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_16bit_encoder_gradients_per_tensor_against_exact_f32_at_224(dtname):
    """ResNet-50 encoder, 16 images of 224 x 224 in one pass (one view: two views in one batch need 128 images each for
    the 7 x 7 maps to fill whole tiles -- tests/test_config_gpu.py runs that form at B = 256), train mode, one upstream
    gradient for all modes.  Per parameter tensor: cosine and norm ratio of
    the 16-bit gradient against the exact-f32 mode's; per stage the WORST tensor is what is asserted.  A gradient that is
    wrong in one stage (a shifted tap, a halo row read from the neighbouring image, a mask bit off by one channel) drops
    that stage's cosine to ~0; rounding noise does not: measured values in the assertion's comment."""
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    g = torch.Generator().manual_seed(17)
    N, views = 16, 1
    # image-like inputs (per-image colour offset + a smooth pattern + noise): N(0,1) noise images are the worst-conditioned
    # input there is for 53 train-mode BatchNorms
    base = torch.nn.functional.interpolate(torch.randn(N, 3, 7, 7, generator=g), size=(224, 224), mode="bilinear")
    x = (base * 1.5 + torch.randn(N, 3, 1, 1, generator=g) + 0.5 * torch.randn(N, 3, 224, 224, generator=g)).to(DEV)
    dfeat = (torch.randn(N, 2048, generator=g) * 1e-2).to(DEV)
    f32, g32 = _encoder_grads(torch.float32, x, dfeat, views)
    f16, g16 = _encoder_grads(dt, x, dfeat, views)
    ferr = float((f16 - f32).norm() / f32.norm())
    worst = {}
    for n, ref in g32.items():
        got = g16[n]
        cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-300))
        ratio = float(got.norm() / (ref.norm() + 1e-300))
        s = _stage(n)
        w = worst.setdefault(s, [1.0, n, 1.0, 1.0])
        if cos < w[0]:
            w[0], w[1] = cos, n
        w[2], w[3] = min(w[2], ratio), max(w[3], ratio)
    allg = lambda d: torch.cat([v.flatten() for v in d.values()])
    a, b = allg(g16), allg(g32)
    whole = float((a * b).sum() / (a.norm() * b.norm()))
    print(f"{dtname} encoder at 224: features rel. error {ferr:.4f}; whole-gradient cosine {whole:.4f}; per stage "
          + "; ".join(f"{s}: worst cosine {w[0]:.4f} ({w[1]}), |g| ratio {w[2]:.3f}..{w[3]:.3f}" for s, w in sorted(worst.items())))
    lim = BOUNDS[dtname]
    assert ferr < lim["features"], ferr
    assert whole > lim["whole"], whole
    for s, w in worst.items():
        assert w[0] > lim["stage"][s], (s, w)
        assert lim["ratio"][0] < w[2] and w[3] < lim["ratio"][1], (s, w)


def _eval_grads(dtype, state, x):
    import resnet
    m = resnet.resnet50(weights=None)
    m.fc = torch.nn.Identity()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    m.sm3_dtype = dtype
    m.to(DEV).eval()
    f = m(x.to(DEV))
    (f.double() ** 2).sum().backward()
    torch.cuda.synchronize()
    return f.detach().double().cpu(), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_16bit_eval_mode_encoder_gradients_per_tensor_at_224(dtname):
    """The same comparison with FROZEN BatchNorm statistics (module.eval() with trainable parameters: the frozen-encoder
    modes of tools/backbone_eval.py / mlc_train.py, src/models/simclr.py:393-396).  Without batch statistics in the loop the
    backward pass is well conditioned (the exact-f32 path holds 2e-3 per tensor against fp64 here:
    tests/test_config_gpu.py::test_eval_mode_backward_through_an_encoder), so THIS is where a 16-bit gradient can be pinned
    tensor by tensor: every convolution's data- and weight-gradient kernel at the network's real map sizes (halo-resident
    and strided 3x3, pointwise 1x1, the direct stem), bf16 and fp16 against exact f32 on the same weights and images."""
    from oracle import procedural
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    state = procedural.make_state_dict(procedural.resnet50_spec(""), seed=17)
    x = torch.from_numpy(procedural.make_images(16, 224, 17, "derm0"))
    f32, g32 = _eval_grads(torch.float32, state, x)
    f16, g16 = _eval_grads(dt, state, x)
    ferr = float((f16 - f32).norm() / f32.norm())
    worst = {}
    for n, ref in g32.items():
        got = g16[n]
        cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-300))
        ratio = float(got.norm() / (ref.norm() + 1e-300))
        w = worst.setdefault(_stage(n), [1.0, n, 1.0, 1.0])
        if cos < w[0]:
            w[0], w[1] = cos, n
        w[2], w[3] = min(w[2], ratio), max(w[3], ratio)
    print(f"{dtname} eval-mode encoder at 224: features rel. error {ferr:.5f}; per stage "
          + "; ".join(f"{s}: worst cosine {w[0]:.5f} ({w[1]}), |g| ratio {w[2]:.4f}..{w[3]:.4f}" for s, w in sorted(worst.items())))
    lim = EVAL_BOUNDS[dtname]
    assert ferr < lim["features"], ferr
    for s, w in worst.items():
        assert w[0] > lim["cos"], (s, w)
        assert lim["ratio"][0] < w[2] and w[3] < lim["ratio"][1], (s, w)


# measured (r5d, MI355X): bf16 features 5.0e-3, worst tensor per stage 0.99979 .. 0.99992, |g| ratio 0.989 .. 1.000;
# fp16 6.2e-4, 0.99999 .. 1.00000, 0.998 .. 1.001.  Bounds: about 5x the measured distance from 1 -- a tap shifted by one
# pixel or one stage's mask off by a channel leaves its tensors at cosine < 0.9.
EVAL_BOUNDS = {"bf16": {"features": 2e-2, "cos": 0.999, "ratio": (0.97, 1.02)},
               "f16": {"features": 3e-3, "cos": 0.9999, "ratio": (0.995, 1.005)}}

# TRAIN mode, measured (r5d): fp16 features 1.8e-2, whole-gradient cosine 0.714, worst tensor per stage 0.665 (layer1) /
# 0.676 / 0.717 / 0.805 (layer4) / 0.710 (stem), |g| ratio 0.85 .. 1.21 -- floors at about 3/4 of the measured values: an
# addressing bug in ONE stage drops that stage's worst tensor to ~0 while fp16's rounding noise does not, and bf16 and fp16
# run the SAME kernel templates (the MFMA builtin and the conversions are all that differs).
# bf16: features 9.3e-2, |g| ratio 0.82 .. 1.23 -- and NO direction: whole-gradient cosine 0.138, worst tensors -0.02 ..
# 0.35, although the loss is not in the loop.  Batch statistics through 53 BatchNorms amplify 8-bit significands to O(1)
# direction errors at random init (PyTorch's own fp32 run is 1.6 % off its fp64 run here: tests/test_config_gpu.py::
# test_train_mode_encoder_gradients_...), so for bf16 the train-mode test pins magnitudes only; its direction is pinned
# where it is defined -- frozen statistics, above -- and its training behaviour by T2 (tests/test_round3_gpu.py).
_ST = ("stem", "layer1", "layer2", "layer3", "layer4")
BOUNDS = {
    "bf16": {"features": 0.2, "whole": -1.0, "stage": {s: -1.0 for s in _ST}, "ratio": (0.6, 1.6)},
    "f16": {"features": 0.04, "whole": 0.6, "stage": {s: 0.5 for s in _ST}, "ratio": (0.7, 1.4)},
}


# ---- VERDICT r4 item 2, forward half: bn1's apply + ReLU inside conv2's halo-resident A image (sm3_conv3x3_bnin) ---------
@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_conv3x3_bnin_equals_bn_act_then_conv(dtname):
    """conv2 reading conv1's RAW output -- BatchNorm affine + ReLU applied to the staged input image in LDS, activation and
    ReLU bits written on the side -- against the two launches it replaces (sm3_bn_act, then sm3_conv_gather_gemm): the
    convolution output, its BatchNorm partial sums, the activation and the mask bit for bit.  Geometries: every stage's
    width, one and two views (scale / shift per view), images that end inside a tile, M not a multiple of 128; launches
    the halo kernel would not take (small grids, a W that does not fit, stride 2) are refused by sm3_conv3x3_bnin_ok."""
    import math
    from sm3hip import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    code = ops.dtype_code(dt)
    D = torch.device(DEV)
    cases = [(40, 56, 56, 64, 64, 1), (64, 28, 28, 128, 128, 2), (200, 14, 14, 256, 256, 1), (768, 7, 7, 512, 512, 2),
             (42, 30, 28, 128, 128, 1), (256, 14, 14, 256, 256, 2)]
    for ci_, (N, H, W, Ci, Co, V) in enumerate(cases):
        g = torch.Generator().manual_seed(50 + ci_)
        d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
        assert ops.conv3x3_bnin_ok(d, V), (N, H, W, Ci, Co, V)
        M = N * H * W
        x = torch.randn(M, Ci, generator=g).to(dt).to(D)
        w = (torch.randn(Co, 9 * Ci, generator=g) / math.sqrt(9 * Ci)).to(dt).to(D)
        scale = (torch.rand(V * Ci, generator=g) + 0.5).to(D)
        shift = (torch.randn(V * Ci, generator=g) * 0.5).to(D)
        # two launches
        act = torch.empty(M, Ci, dtype=dt, device=D)
        mask = torch.empty(M * Ci // 8, dtype=torch.uint8, device=D)
        ops.bn_act(code, x, scale, shift, None, True, act, M // V, Ci, mask=mask, views=V)
        y = torch.empty(M, Co, dtype=dt, device=D)
        part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=D)
        ops.conv_gemm(d, act, w, y, None, part)
        # one launch
        act2 = torch.full((M, Ci), float("nan"), dtype=dt, device=D)
        mask2 = torch.full((M * Ci // 8,), 0xAA, dtype=torch.uint8, device=D)
        y2 = torch.empty(M, Co, dtype=dt, device=D)
        part2 = torch.zeros_like(part)
        ops.conv3x3_bnin(d, x, scale, shift, act2, mask2, w, y2, part2, views=V)
        torch.cuda.synchronize()
        what = (dtname, N, H, W, Ci, Co, V)
        assert torch.equal(act2, act), ("activation",) + what
        assert torch.equal(mask2, mask), ("relu bits",) + what
        assert torch.equal(y2, y), ("conv output",) + what
        assert torch.equal(part2, part), ("bn partial sums",) + what
    # refused: a grid of at most 256 workgroups (the deep kernel's), stride 2, a 1x1, the exact-f32 type
    assert not ops.conv3x3_bnin_ok(ops.fwd_desc(code, 2, 14, 14, 256, 256, 3, 1, 1))
    assert not ops.conv3x3_bnin_ok(ops.fwd_desc(code, 64, 28, 28, 128, 128, 3, 2, 1))
    assert not ops.conv3x3_bnin_ok(ops.fwd_desc(code, 64, 28, 28, 128, 128, 1, 1, 0))
    assert not ops.conv3x3_bnin_ok(ops.fwd_desc(code, 64, 112, 112, 64, 64, 3, 1, 1))          # W = 112: the image does not fit
    assert not ops.conv3x3_bnin_ok(ops.fwd_desc(ops.dtype_code(torch.float32), 64, 28, 28, 128, 128, 3, 1, 1))


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_step_with_and_without_bnin_is_the_same_step(dtname):
    """A whole SM3 step (B = 32 pairs of 224 x 224, both views in one batch) with bn1's apply fused into conv2
    (SM3_CONV_BNIN=1; measured slower, so opt-in: DESIGN.md 3.6.5) and with the separate apply pass: the forward is bit-identical -- same loss, same running
    statistics -- and the gradients agree to the float-atomic noise of the weight-gradient sums."""
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    B, S = 32, 224
    g = torch.Generator(device=DEV).manual_seed(9)
    derm = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    clinic = [torch.randn(B, 3, S, S, device=DEV, generator=g) for _ in range(2)]
    torch.manual_seed(9)
    init = {k: v.clone() for k, v in SimCLRSkinV32("resnet50", None, 128, 0.1).state_dict().items()}
    runs = {}
    for on in (True, False):
        model = SimCLRSkinV32("resnet50", None, 128, 0.1)
        model.load_state_dict(init)
        model.sm3_dtype = dt
        model.to(DEV)
        tr = SM3Trainer(model, lr=0.0, init_scale=1024.0)
        eng = tr._engine()
        eng.bnin = on
        calls = []
        from sm3hip import ops
        real = ops.conv3x3_bnin
        ops.conv3x3_bnin = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            loss = float(tr.step(derm, clinic))
        finally:
            ops.conv3x3_bnin = real
        torch.cuda.synchronize()
        runs[on] = (loss, eng.store.flat_g.double().clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}, len(calls))
        del tr, eng, model
        torch.cuda.empty_cache()
    # the stride-1 3x3 units of layers 1 and 2 (3 + 3 per encoder, two branches, both views per launch); at this batch the
    # grids of layers 3 and 4 stay below 257 workgroups and keep the two-pass form
    assert runs[True][3] == 2 * 6 and runs[False][3] == 0
    for k, v in runs[True][2].items():
        assert torch.equal(v, runs[False][2][k]), k
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    rel = float((runs[True][1] - runs[False][1]).norm() / runs[False][1].norm())
    assert rel < 1e-5, rel
