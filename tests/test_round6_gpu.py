"""Round-6 cases (GPU), through the C ABI: a training step is a FUNCTION OF ITS INPUTS.

  * sm3_conv_wgrad_det (plain-store split-K slabs + one fixed-order sum) against fp64 `conv2d_weight` on the shapes of
    test_kernels_gpu, bit-identical over repeats with and without a second stream keeping the chip busy, and within fp32
    rounding of the float-atomic form it replaces;
  * sm3_slab_reduce against an fp64 sum, store and accumulate forms;
  * the stem weight gradient with slabs against the atomic form and over repeats;
  * three optimizer steps of the whole model, twice from the same state, in the three arithmetic modes: every parameter
    bit, every AdamW moment bit and all three losses equal (reference: tools/backbone_train.py:98-127 is a deterministic
    program on the reference's CPU path, SURVEY.md 8c "fp32 CPU reruns are bitwise reproducible").
"""
import math

import pytest
import torch
import torch.nn.functional as F

from test_config_gpu import DEV, _batch, _build
from test_kernels_gpu import CONV_CASES, DTYPES3, IDS3, _ops, dev, nhwc, rnd, tol

pytestmark = pytest.mark.gpu


def _busy(n=6):
    """A second stream that keeps every CU busy while the kernels under test run (perturbs the order in which workgroups
    are dispatched and retire -- what an order-dependent sum would be sensitive to)."""
    st = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(st):
        for _ in range(n):
            a = (a @ a).clamp_(-1, 1)
    return st, a


DET_CASES = CONV_CASES + [
    # many pixel slices: layer-1 shapes of the step at B = 32 / 64 x 64 and a 3x3 with 36 tiles
    (64, 64, 64, 56, 56, 1, 1, 0),
    (32, 64, 64, 56, 56, 3, 1, 1),
    (64, 256, 128, 28, 28, 1, 1, 0),
    (32, 256, 256, 14, 14, 3, 1, 1),
    # layer-3 / layer-4 dense shapes: the 128 x 256 weight-gradient tile (Co x Ci >= 256 x 1024, Ci a multiple of 256)
    (8, 1024, 256, 14, 14, 1, 1, 0),
    (8, 256, 1024, 14, 14, 1, 1, 0),
    (6, 512, 2048, 7, 7, 1, 1, 0),
]


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("case", DET_CASES)
def test_conv_wgrad_det_is_exact_and_a_function_of_its_inputs(case, dt):
    ops = _ops()
    N, Ci, Co, H, W, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Ci, H, W, generator=g)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    code = ops.dtype_code(dt)
    xr, dyr = rnd(x, dt), rnd(dy, dt)
    ref_dw = torch.nn.grad.conv2d_weight(xr.double(), (Co, Ci, k, k), dyr.double(), stride=s, padding=p)
    xd, dyd = nhwc(x, dt), nhwc(dy, dt)
    d = ops.fwd_desc(code, N, H, W, Ci, Co, k, s, p)
    n = Co * k * k * Ci
    cap = ops.wgrad_det_cap(n)
    slabs = torch.full((cap * n,), float("nan"), device=dev())  # whatever is read must have been written by this launch
    base = torch.randn(Co, k * k * Ci, generator=g).to(dev())   # the gradient buffer already holds something: dw += ...
    outs = []
    for rep in range(4):
        dw = base.clone()
        busy = _busy() if rep % 2 else None
        ops.conv_wgrad_det(d, xd, dyd, dw, slabs, cap)
        torch.cuda.synchronize()
        del busy
        outs.append(dw)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    got = (outs[0] - base).cpu().reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
    sc = ref_dw.abs().max().item()
    assert (got.double() - ref_dw).abs().max().item() < tol(dt, sc) * 2
    # the float-atomic form computes the same sum in another order
    dwa = base.clone()
    ops.conv_wgrad(d, xd, dyd, dwa)
    torch.cuda.synchronize()
    assert (dwa - outs[0]).abs().max().item() < 1e-5 * max(sc, 1.0) * math.sqrt(N * Ho * Wo / 64 + 1)


@pytest.mark.parametrize("nslabs,n", [(1, 64), (3, 4096), (8, 9408), (37, 128 * 576), (256, 4096), (768, 9408)])
def test_slab_reduce(nslabs, n):
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(nslabs * 7 + n)
    slabs = torch.randn(nslabs, n, device=DEV, generator=g)
    out = torch.full((n + 4,), 7.0, device=DEV)
    ops.slab_reduce(slabs.view(-1), nslabs, n, out, accumulate=False)
    ref = slabs.double().sum(0)
    assert (out[:n].double() - ref).abs().max().item() < 1e-5 * math.sqrt(nslabs) * 4
    assert bool((out[n:] == 7.0).all())                      # nothing past n is touched
    first = out[:n].clone()
    ops.slab_reduce(slabs.view(-1), nslabs, n, out, accumulate=True)
    assert torch.equal(out[:n], first + first)               # out + (the same fixed-order sum)
    again = torch.empty(n, device=DEV)
    ops.slab_reduce(slabs.view(-1), nslabs, n, again, accumulate=False)
    assert torch.equal(again, first)


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("views", [1, 2])
def test_stem_wgrad_with_slabs_equals_the_atomic_form_and_repeats_bit_for_bit(dt, views):
    ops = _ops()
    code = ops.dtype_code(dt)
    N, H, W = 8 * views, 64, 64
    Ho, Wo = H // 2, W // 2
    g = torch.Generator(device=DEV).manual_seed(11 + views)
    x = torch.randn(N, 3, H, W, device=DEV, generator=g)
    dz = torch.randn(N * Ho * Wo, 64, device=DEV, generator=g).to(dt)
    xo = torch.randn(N * Ho * Wo, 64, device=DEV, generator=g).to(dt)
    mean = torch.randn(views * 64, device=DEV, generator=g) * 0.1
    invstd = torch.rand(views * 64, device=DEV, generator=g) + 0.5
    gamma = torch.rand(64, device=DEV, generator=g) + 0.5
    gsums = torch.randn(views * 128, device=DEV, generator=g, dtype=torch.float64)
    count = float(N // views * Ho * Wo)
    slabs = torch.full((ops.STEM_WGRAD_SLABS * 64 * 147,), float("nan"), device=DEV)

    def run(use_slabs, busy):
        dw = torch.ones(64 * 147, device=DEV)
        dg, db = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
        b = _busy() if busy else None
        ops.stem_wgrad_bn(code, x, dz, xo, mean, invstd, gamma, gsums, count, gsums, dg, db, dw, views=views,
                          slabs=slabs if use_slabs else None)
        torch.cuda.synchronize()
        del b
        return dw, dg, db

    ref = run(False, False)
    outs = [run(True, i % 2 == 1) for i in range(4)]
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    sc = float((ref[0] - 1).abs().max())
    assert float((outs[0][0] - ref[0]).abs().max()) < 2e-5 * max(sc, 1.0)
    assert torch.equal(outs[0][1], ref[1]) and torch.equal(outs[0][2], ref[2])   # d(gamma), d(beta): one thread, view order


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
def test_three_training_steps_are_a_function_of_their_inputs(dt):
    """Two trainers from one state, three AdamW steps each on three different batches (the second one beside a stream that
    keeps the chip busy): losses, parameters, moments and gradients equal to the last bit."""
    from sm3hip.trainer import SM3Trainer
    B, S = 32, 64
    batches = [_batch(B, S, 60 + i) for i in range(3)]
    runs = []
    for rep in range(2):
        model = _build(17, dt)
        tr = SM3Trainer(model, lr=1e-3, weight_decay=5e-2, eps=1e-5, style=0, init_scale=256.0)
        eng = tr._engine()
        assert eng.det_wgrad
        busy = _busy(40) if rep else None
        losses = [float(tr.step(*b)) for b in batches]
        torch.cuda.synchronize()
        del busy
        if dt == torch.float16:
            assert tr.steps_taken() == 3
        st = eng.store
        runs.append((losses, st.flat_p.clone(), tr.m.clone(), tr.v.clone(), st.flat_g.clone(),
                     {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
        del tr, eng, model
        torch.cuda.empty_cache()
    a, b = runs
    assert a[0] == b[0], (a[0], b[0])
    for i, name in ((1, "parameters"), (2, "exp_avg"), (3, "exp_avg_sq"), (4, "gradients")):
        assert torch.equal(a[i], b[i]), (name, int((a[i] != b[i]).sum()))
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k
    assert all(l == l for l in a[0]) and a[0][0] != a[0][1]


STEM16_CASES = [(4, 64, 64), (2, 224, 224), (3, 96, 80), (2, 70, 54), (2, 448, 448), (1, 37, 301)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", STEM16_CASES)
def test_stem_kernels_on_the_16bit_image_equal_the_fp32_image_kernels_bit_for_bit(case, dt):
    """sm3_stem_image_prep + sm3_stem_conv_fwd16 / sm3_stem_wgrad_bn16 (images rounded once, rows staged by LDS-DMA) against
    sm3_stem_conv_fwd / sm3_stem_wgrad_bn on the fp32 images (rounded per staged tile): the same round-to-nearest-even of
    the same values, the same MFMA sequence -- outputs, BatchNorm partial rows and the weight gradient are EQUAL.  Two views
    come from two tensors (no concatenated copy).  The padded image itself is checked against torch."""
    ops = _ops()
    code = ops.dtype_code(dt)
    B, H, W = case
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + H + W)
    views = [torch.randn(B, 3, H, W, device=DEV, generator=g) for _ in range(2)]
    w_master = (torch.randn(64, 147, device=DEV, generator=g) / 12.0).contiguous()
    w_stem = torch.empty(64 * ops.STEM_KDIRECT, dtype=dt, device=DEV)
    ops.stem_weight_prep(code, w_master, w_stem, None)
    for V in (1, 2):
        xs = views[:V]
        N = V * B
        img = ops.stem_image_prep(code, xs)
        Wp = ops.stem_image_cols(W)
        want = torch.zeros(N, 3, H, Wp, dtype=dt, device=DEV)
        want[..., 3:3 + W] = torch.cat(xs, 0).to(dt)
        assert torch.equal(img.t, want)
        xcat = torch.cat(xs, 0).contiguous()
        prow = ops.stem_partial_rows(N, H, W)
        y_a, y_b = (torch.empty(N * Ho * Wo, 64, dtype=dt, device=DEV) for _ in range(2))
        p_a, p_b = (torch.full((prow * 2 * 64,), float("nan"), device=DEV) for _ in range(2))
        ops.stem_conv_fwd(code, xcat, w_stem, y_a, p_a)
        ops.stem_conv_fwd16(code, img, w_stem, y_b, p_b)
        torch.cuda.synchronize()
        assert torch.equal(y_a, y_b) and torch.equal(p_a, p_b)
        # weight gradient with bn1's backward apply in the operand load
        dz = torch.randn(N * Ho * Wo, 64, device=DEV, generator=g).to(dt)
        mean = torch.randn(V * 64, device=DEV, generator=g) * 0.1
        invstd = torch.rand(V * 64, device=DEV, generator=g) + 0.5
        gamma = torch.rand(64, device=DEV, generator=g) + 0.5
        gsums = torch.randn(V * 128, device=DEV, generator=g, dtype=torch.float64)
        count = float(B * Ho * Wo)
        slabs = torch.full((ops.STEM_WGRAD_SLABS * 64 * 147,), float("nan"), device=DEV)
        outs = []
        for fn, im in ((ops.stem_wgrad_bn, xcat), (ops.stem_wgrad_bn16, img)):
            for use_slabs in (True, False):
                dw = torch.zeros(64 * 147, device=DEV)
                dg, db = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
                fn(code, im, dz, y_a, mean, invstd, gamma, gsums, count, gsums, dg, db, dw, views=V,
                   slabs=slabs if use_slabs else None)
                torch.cuda.synchronize()
                outs.append((dw, dg, db))
        assert torch.equal(outs[0][0], outs[2][0])                       # fixed-order forms: equal
        for o in outs[1:]:
            assert torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
            sc = float(outs[0][0].abs().max())
            assert float((o[0] - outs[0][0]).abs().max()) < 2e-5 * max(sc, 1.0)


# ---- nine-tap owner weight gradient (csrc/conv_wgrad.hip: conv_wgrad9_kernel) -------------------------------------------
WGRAD9_CASES = [
    # N, Ci, Co, H, W: the step's four 3x3 geometries (R = 1 / 4 / 7 / 7 rows per stage), odd maps (R = 1 on a prime H,
    # a 6 x 6 map staged whole), a map whose stage only fits a one-stage ring (W = 112), Co != Ci
    (4, 64, 64, 56, 56), (4, 128, 128, 28, 28), (8, 256, 256, 14, 14), (8, 512, 512, 7, 7),
    (3, 64, 128, 41, 37), (5, 128, 64, 6, 6), (2, 64, 64, 112, 112), (3, 192, 128, 12, 10), (2, 64, 64, 16, 16),
]


def _wgrad9_inputs(case, dt):
    N, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Ci, H, W, generator=g)
    dy = torch.randn(N, Co, H, W, generator=g)
    return x, dy


@pytest.mark.parametrize("dt", DTYPES3[1:], ids=IDS3[1:])
@pytest.mark.parametrize("case", WGRAD9_CASES)
def test_nine_tap_owner_wgrad_against_fp64_and_the_tap_shifted_kernel(case, dt, monkeypatch):
    ops = _ops()
    N, Ci, Co, H, W = case
    x, dy = _wgrad9_inputs(case, dt)
    code = ops.dtype_code(dt)
    ref_dw = torch.nn.grad.conv2d_weight(rnd(x, dt).double(), (Co, Ci, 3, 3), rnd(dy, dt).double(), stride=1, padding=1)
    xd, dyd = nhwc(x, dt), nhwc(dy, dt)
    d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
    n = Co * 9 * Ci
    cap = ops.wgrad_det_cap(n)
    slabs = torch.full((cap * n,), float("nan"), device=dev())
    sc = ref_dw.abs().max().item()
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SM3_WGRAD9", flag)
        dwa = torch.zeros(Co, 9 * Ci, device=dev())
        ops.conv_wgrad(d, xd, dyd, dwa)           # float-atomic form
        dwd = torch.zeros(Co, 9 * Ci, device=dev())
        ops.conv_wgrad_det(d, xd, dyd, dwd, slabs, cap)  # slabs + fixed-order sum
        torch.cuda.synchronize()
        for o in (dwa, dwd):
            got = o.cpu().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)
            assert (got.double() - ref_dw).abs().max().item() < tol(dt, sc) * 2
        res[flag] = (dwa, dwd)
    # same products, another partition of the pixel axis: fp32 reassociation only
    for a, b in zip(res["1"], res["0"]):
        assert (a - b).abs().max().item() < 2e-5 * sc * math.sqrt(N * H * W)


@pytest.mark.parametrize("dt", DTYPES3[1:], ids=IDS3[1:])
@pytest.mark.parametrize("case", [(4, 128, 128, 28, 28), (2, 64, 64, 16, 16), (3, 64, 128, 32, 8)])
def test_nine_tap_owner_wgrad_bits_equal_the_tap_shifted_kernel_on_one_slice(case, dt, monkeypatch):
    """With ONE pixel slice and stages that are whole 16-pixel K blocks (R * W a multiple of 16: R = 4 / 8 / 16 by the
    library's rule here) both kernels feed every accumulator the same 16-pixel groups in the same order; masked and
    zero-filled products are exact zeros: equal bits."""
    ops = _ops()
    N, Ci, Co, H, W = case
    x, dy = _wgrad9_inputs(case, dt)
    code = ops.dtype_code(dt)
    xd, dyd = nhwc(x, dt), nhwc(dy, dt)
    d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
    monkeypatch.setenv("SM3_WGRAD_TARGET_CTAS", "1")
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SM3_WGRAD9", flag)
        dw = torch.zeros(Co, 9 * Ci, device=dev())
        ops.conv_wgrad(d, xd, dyd, dw)
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])


# ---- the step's NT-Xent terms as one launch per phase (csrc/ntxent.hip: sm3_ntxent_fused_batch) ----------------------------
@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("n,R,D,scaled", [(4, 512, 128, False), (4, 64, 128, True), (2, 24, 64, False), (1, 512, 128, True),
                                          (3, 130, 32, False)])
def test_batched_ntxent_terms_equal_the_per_term_calls_bit_for_bit(n, R, D, scaled, dt):
    ops = _ops()
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(n * 1000 + R + D)
    zs = [torch.randn(R, D, generator=g).to(dev()) for _ in range(n)]
    weights = [1.0, 1.0, 0.5, 0.5][:n]
    scale = torch.tensor([1024.0], device=dev()) if scaled else None
    per = ops.ntxent_workspace_floats(R, D)
    # term by term (what the trainer did until round 6)
    loss_ref = torch.zeros(1, device=dev())
    dz_ref = [torch.empty(R, D, dtype=dt, device=dev()) for _ in range(n)]
    ws1 = torch.empty(per, device=dev())
    for z, w, d in zip(zs, weights, dz_ref):
        ops.ntxent_fused(code, z, 0.1, w, ws1, loss_ref, d, dz_scale=scale)
    # one launch per phase
    loss = torch.zeros(1, device=dev())
    dz = [torch.full((R, D), float("nan"), dtype=dt, device=dev()) for _ in range(n)]
    ws = torch.empty(ops.ntxent_batch_workspace_floats(n, R, D), device=dev())
    assert ops.ntxent_fused_batch(code, zs, 0.1, weights, ws, loss, dz, dz_scale=scale)
    torch.cuda.synchronize()
    assert torch.equal(loss, loss_ref) and float(loss) > 0
    for a, b in zip(dz, dz_ref):
        assert torch.equal(a, b)
