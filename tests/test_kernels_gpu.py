"""Kernel-level parity (GPU): every C-ABI entry point against a plain PyTorch CPU fp32/fp64 reference of
the same op on the same seeded inputs.  bf16 cases feed the reference the bf16-rounded inputs, so only
accumulation order and the final rounding differ."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
DTYPES3 = [torch.float32, torch.bfloat16, torch.float16]  # + the reference's AMP storage type (round 2)
IDS3 = ["f32", "bf16", "f16"]


def _ops():
    from sm3hip import ops
    return ops


def dev():
    return torch.device("cuda:0")


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(dt).to(dev())


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(x, dt):
    return x.to(dt).float()


def tol(dt, scale):
    return {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}[dt] * scale


CONV_CASES = [
    # N, Ci, Co, H, W, k, s, p
    (2, 64, 64, 14, 14, 1, 1, 0),
    (2, 64, 256, 9, 7, 1, 1, 0),
    (3, 128, 128, 12, 10, 3, 1, 1),
    (2, 64, 64, 13, 11, 3, 2, 1),
    (2, 256, 512, 10, 10, 1, 2, 0),
    (1, 512, 128, 7, 7, 1, 1, 0),
    (5, 128, 192, 6, 6, 3, 1, 1),
    # many tiles per launch, M tail
    (16, 64, 256, 56, 56, 1, 1, 0),
    (13, 128, 256, 41, 37, 3, 1, 1),
    (9, 256, 384, 50, 46, 1, 2, 0),
    # 3x3 / stride 1 with many row tiles: four channel chunks, 64-column tiles on a wide map, an M tail on an odd map

    (8, 256, 256, 14, 14, 3, 1, 1),
    (2, 64, 64, 56, 56, 3, 1, 1),
    (3, 128, 320, 29, 23, 3, 1, 1),
]


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dt):
    ops = _ops()
    N, Ci, Co, H, W, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k)
    code = ops.dtype_code(dt)
    xr, wr = rnd(x, dt), rnd(w, dt)
    ref = F.conv2d(xr.double(), wr.double(), stride=s, padding=p)
    Ho, Wo = ref.shape[2:]

    xd = nhwc(x, dt)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()  # [Co][k][k][Ci]
    wd = w_ohwi.to(dt).to(dev())
    d = ops.fwd_desc(code, N, H, W, Ci, Co, k, s, p)
    assert (d.Ho, d.Wo) == (Ho, Wo)
    y = torch.empty(N, Ho, Wo, Co, dtype=dt, device=dev())
    prow = ops.conv_partial_rows(d)
    partials = torch.full((prow, 2, Co), float("nan"), device=dev())
    ops.conv_gemm(d, xd, wd, y, None, partials)
    torch.cuda.synchronize()
    got = from_nhwc(y)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < tol(dt, scale)
    # BN partial sums are sums of the stored values
    yv = y.float().reshape(-1, Co).double()
    ps = partials.double().sum(0).cpu()
    assert torch.allclose(ps[0], yv.sum(0).cpu(), rtol=1e-4, atol=1e-3 * scale)
    assert torch.allclose(ps[1], (yv * yv).sum(0).cpu(), rtol=1e-4, atol=1e-3 * scale * scale)

    # ---- data gradient ----
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    dyr = rnd(dy, dt)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), wr.double(), dyr.double(), stride=s, padding=p)
    dyd = nhwc(dy, dt)
    # transposed filter bank via the library's own weight_prep
    w_master = w_ohwi.to(dev())
    w_dg = torch.empty(Ci, k * k, Co, dtype=dt, device=dev())
    w_f = torch.empty(Co, k * k * Ci, dtype=dt, device=dev())
    ops.weight_prep(code, w_master, Co, k * k, Ci, w_f, k * k * Ci, w_dg)
    torch.cuda.synchronize()
    assert torch.equal(w_f.cpu().reshape(-1), wd.cpu().reshape(-1))
    descs, full = ops.dgrad_descs(code, N, H, W, Ci, Co, k, s, p)
    addend = torch.randn(N, H, W, Ci, generator=g).to(dt)
    dx = addend.clone().to(dev()) if not full else torch.empty(N, H, W, Ci, dtype=dt, device=dev())
    add_dev = addend.to(dev())
    for dd in descs:
        ops.conv_gemm(dd, dyd, w_dg, dx, dx if not full else add_dev, None)
    torch.cuda.synchronize()
    want = ref_dx + addend.float().permute(0, 3, 1, 2).double()
    sc = want.abs().max().item()
    assert (from_nhwc(dx).double() - want).abs().max().item() < tol(dt, sc) * 2

    # ---- weight gradient (accumulates) ----
    ref_dw = torch.nn.grad.conv2d_weight(xr.double(), (Co, Ci, k, k), dyr.double(), stride=s, padding=p)
    dw = torch.ones(Co, k * k * Ci, device=dev())
    ops.conv_wgrad(d, xd, dyd, dw)
    torch.cuda.synchronize()
    got_dw = (dw.cpu() - 1).reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
    sc = ref_dw.abs().max().item()
    assert (got_dw.double() - ref_dw).abs().max().item() < tol(dt, sc) * 2


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("case", [(2, 64, 128, 9, 11, 1, 1, 0), (3, 128, 64, 10, 8, 3, 1, 1), (2, 64, 192, 13, 11, 3, 2, 1),
                                  (12, 256, 128, 57, 50, 1, 1, 0), (8, 128, 64, 66, 62, 3, 2, 1),
                                  # 3x3 stride 1 with many row tiles, 64- and 128-column tiles, M tails
                                  (6, 64, 128, 30, 28, 3, 1, 1), (2, 128, 128, 56, 56, 3, 1, 1), (5, 128, 192, 19, 21, 3, 1, 1)])
def test_dgrad_with_fused_bn_backward_phase1(case, dt):
    """sm3_conv_dgrad_bnfuse == sm3_conv_gather_gemm followed by sm3_bn_bwd_reduce (dz bit-exact, sums equal)."""
    ops = _ops()
    N, Ci, Co, H, W, k, s, p = case   # forward conv Ci -> Co; its data gradient has Ci output channels
    code = ops.dtype_code(dt)
    E = 4 if dt == torch.float32 else 8
    g = torch.Generator().manual_seed(sum(case))
    D = dev()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, Ho, Wo, Co, generator=g).to(dt).to(D)
    w_dg = (torch.randn(Ci, k * k, Co, generator=g) / math.sqrt(Co * k * k)).to(dt).to(D)
    addend = torch.randn(N, H, W, Ci, generator=g).to(dt).to(D)
    bn_x = torch.randn(N * H * W, Ci, generator=g).to(dt).to(D)             # producer BN's input
    mask = torch.randint(0, 256, (N * H * W * Ci // E,), generator=g, dtype=torch.uint8).to(D)
    if E == 4:
        mask &= 0x0F
    mean, invstd = torch.randn(Ci, generator=g).to(D), (torch.rand(Ci, generator=g) + 0.5).to(D)
    descs, full = ops.dgrad_descs(code, N, H, W, Ci, Co, k, s, p)
    assert full
    rows = N * H * W
    # reference: plain data gradient, then the standalone phase-1 kernel
    ref = torch.empty(rows, Ci, dtype=dt, device=D)
    for dd in descs:
        ops.conv_gemm(dd, dy, w_dg, ref, addend, None)
    prow_ref = ops.bn_bwd_partial_rows(rows, Ci)
    part_ref = torch.zeros(prow_ref, 2, Ci, device=D)
    ops.bn_bwd_reduce(code, ref, None, bn_x, mean, invstd, ref, rows, Ci, part_ref, mask=mask)
    # fused
    out = torch.empty(rows, Ci, dtype=dt, device=D)
    total = sum(ops.conv_partial_rows(dd) for dd in descs)
    part = torch.full((total, 2, Ci), float("nan"), device=D)
    off = 0
    for dd in descs:
        off += ops.conv_dgrad_bnfuse(dd, dy, w_dg, out, addend, mask, bn_x, mean, invstd, part, off)
    torch.cuda.synchronize()
    assert off == total
    assert torch.equal(out, ref)
    a, b = part.double().sum(0).cpu(), part_ref.double().sum(0).cpu()
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * float(b.abs().max()))


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
def test_linear_as_conv(dt):
    ops = _ops()
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(5)
    M, K, Nn = 24, 2048, 128
    x = torch.randn(M, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    ref = rnd(x, dt).double() @ rnd(w, dt).double().t()
    d = ops.fwd_desc(code, M, 1, 1, K, Nn, 1, 1, 0)
    y = torch.empty(M, Nn, dtype=dt, device=dev())
    ops.conv_gemm(d, x.to(dt).to(dev()), w.to(dt).to(dev()), y, None, None)
    torch.cuda.synchronize()
    assert (y.float().cpu().double() - ref).abs().max().item() < tol(dt, ref.abs().max().item())


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("case", ["linear_256x2048x2048", "linear_512x2048x128", "conv3x3_2x14x14x256", "ragged_200x1024x192"])
def test_deep_pipeline_equals_two_stage(case, dt, monkeypatch):
    """Launches of at most one workgroup per CU with >= 6 K-steps run the 4-stage K-loop (3 DMA stages in flight) on
    64-column tiles; same MFMA order, so output AND BatchNorm partial sums are bit-identical to the 2-stage kernel
    (SM3_CONV_DEEP=0), and both match fp64 on the rounded operands."""
    ops = _ops()
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(11)
    if case.startswith("conv3x3"):
        N, H, Ci, Co, k, pad = 2, 14, 256, 256, 3, 1
    else:
        M, Ci, Co = {"linear_256x2048x2048": (256, 2048, 2048), "linear_512x2048x128": (512, 2048, 128),
                     "ragged_200x1024x192": (200, 1024, 192)}[case]
        N, H, k, pad = M, 1, 1, 0
    x = rnd(torch.randn(N, Ci, H, H, generator=g), dt)
    w = rnd(torch.randn(Co, Ci, k, k, generator=g) / math.sqrt(Ci * k * k), dt)
    ref = F.conv2d(x.double(), w.double(), padding=pad)
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, pad)
    xd = nhwc(x, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev())
    prow = ops.conv_partial_rows(d)
    outs = {}
    monkeypatch.setenv("SM3_CONV_HALO", "0")  # (the halo-resident 3x3 variant sums the K-steps in another order)
    for deep in ("1", "0"):
        monkeypatch.setenv("SM3_CONV_DEEP", deep)
        y = torch.empty(N * H * H, Co, dtype=dt, device=dev())
        part = torch.full((prow, 2, Co), float("nan"), device=dev())
        ops.conv_gemm(d, xd, wd, y, None, part)
        y2 = torch.empty_like(y)
        addend = torch.zeros_like(y)
        ops.conv_gemm(d, xd, wd, y2, addend, None)     # general epilogue
        torch.cuda.synchronize()
        outs[deep] = (y.clone(), part.clone(), y2.clone())
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)
    y = outs["1"][0].float().cpu().double().reshape(N, H, H, Co).permute(0, 3, 1, 2)
    assert (y - ref).abs().max().item() < tol(dt, ref.abs().max().item())
    if dt != torch.float32:  # lean vs general epilogue: same rounding of the same accumulators
        assert torch.equal(outs["1"][0], outs["1"][2])
    stored = outs["1"][0].float().cpu().double()
    assert torch.allclose(outs["1"][1].cpu().double().sum(0)[0], stored.sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("shape", [(2 * 9 * 9, 64), (300, 256), (7, 2048), (1000, 128)])
def test_bn_train_forward_backward(shape, dt):
    ops = _ops()
    code = ops.dtype_code(dt)
    rows, Cn = shape
    g = torch.Generator().manual_seed(rows)
    x = rnd(torch.randn(rows, Cn, generator=g) * 2 + 0.5, dt)
    res = rnd(torch.randn(rows, Cn, generator=g), dt)
    gamma = torch.rand(Cn, generator=g) + 0.5
    beta = torch.randn(Cn, generator=g) * 0.1
    rm, rv = torch.zeros(Cn), torch.ones(Cn)
    xd64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rm.double(), rv.double()
    ref_bn = F.batch_norm(xd64, rm64, rv64, g64, b64, True, 0.1, 1e-5)
    ref_y = F.relu(ref_bn + res.double())

    # forward: partial sums (one row per 128-row block, as the conv epilogue would write) -> reduce -> finalize
    D = dev()
    xdv = x.to(dt).to(D)
    nblk = (rows + 127) // 128
    partials = torch.zeros(nblk, 2, Cn)
    for b in range(nblk):
        blk = x[b * 128:(b + 1) * 128].double()
        partials[b, 0] = blk.sum(0).float()
        partials[b, 1] = (blk * blk).sum(0).float()
    partials = partials.to(D)
    sums = torch.empty(2 * Cn, dtype=torch.float64, device=D)
    ops.bn_stats_reduce(partials, nblk, Cn, sums)
    scale, shift = torch.empty(Cn, device=D), torch.empty(Cn, device=D)
    mean, invstd = torch.empty(Cn, device=D), torch.empty(Cn, device=D)
    rmd, rvd = rm.clone().to(D), rv.clone().to(D)
    nbt = torch.zeros((), dtype=torch.int64, device=D)
    ops.bn_finalize(sums, rows, Cn, gamma.to(D), beta.to(D), 1e-5, 0.1, rmd, rvd, nbt, scale, shift, mean, invstd)
    y = torch.empty(rows, Cn, dtype=dt, device=D)
    E = 4 if dt == torch.float32 else 8
    mask = torch.empty(rows * Cn // E, dtype=torch.uint8, device=D)
    ops.bn_act(code, xdv, scale, shift, res.to(dt).to(D), True, y, rows, Cn, mask=mask)
    # same statistics through the folded path: stage A only, stage B inside bn_finalize
    ws, groups = ops.bn_stats_reduce(partials, nblk, Cn, None)
    scale2, shift2 = torch.empty(Cn, device=D), torch.empty(Cn, device=D)
    ops.bn_finalize(ws, rows, Cn, gamma.to(D), beta.to(D), 1e-5, 0.1, None, None, None, scale2, shift2, None, None,
                    groups=groups)
    torch.cuda.synchronize()
    assert torch.equal(scale, scale2) and torch.equal(shift, shift2)
    # the mask is exactly (y > 0), bit e of byte v = element 8v+e (4v+e in f32)
    bits = ((mask.cpu().unsqueeze(1) >> torch.arange(E, dtype=torch.uint8)) & 1).reshape(rows, Cn).bool()
    assert torch.equal(bits, y.float().cpu() > 0)
    assert int(nbt) == 1
    sc = ref_y.abs().max().item()
    assert (y.float().cpu().double() - ref_y.detach()).abs().max().item() < tol(dt, sc)
    assert torch.allclose(rmd.cpu(), rm64.float(), atol=1e-5, rtol=1e-4)  # rm64/rv64 updated in place by F.batch_norm
    assert torch.allclose(rvd.cpu(), rv64.float(), atol=1e-5, rtol=1e-4)

    # backward
    dy = rnd(torch.randn(rows, Cn, generator=g), dt)
    ref_y.backward(dy.double())
    prow = ops.bn_bwd_partial_rows(rows, Cn)
    bpart = torch.full((prow, 2, Cn), float("nan"), device=D)
    dz = dy.to(dt).to(D).clone()
    y_ref_dev = ref_y.detach().float().to(dt).to(D)  # mask source: exact reference activations
    ops.bn_bwd_reduce(code, dz, y_ref_dev, xdv, mean, invstd, dz, rows, Cn, bpart)
    # mask-bit variant gives the same dz and partial sums
    ref_mask = torch.zeros(rows * Cn // E, dtype=torch.uint8)
    yb = (ref_y.detach() > 0).reshape(rows * Cn // E, E).to(torch.uint8)
    for e in range(E):
        ref_mask |= yb[:, e] << e
    dz2 = dy.to(dt).to(D).clone()
    bpart2 = torch.full((prow, 2, Cn), float("nan"), device=D)
    ops.bn_bwd_reduce(code, dz2, None, xdv, mean, invstd, dz2, rows, Cn, bpart2, mask=ref_mask.to(D))
    torch.cuda.synchronize()
    assert torch.equal(dz, dz2) and torch.equal(bpart, bpart2)
    lsums = torch.empty(2 * Cn, dtype=torch.float64, device=D)
    ops.bn_stats_reduce(bpart, prow, Cn, lsums)
    dx = torch.empty(rows, Cn, dtype=dt, device=D)
    dgamma, dbeta = torch.ones(Cn, device=D), torch.ones(Cn, device=D)
    ops.bn_bwd_apply(code, dz, xdv, mean, invstd, gamma.to(D), lsums, rows, lsums, dgamma, dbeta, dx, rows, Cn)
    torch.cuda.synchronize()
    mask = (ref_y.detach() > 0).double()
    assert torch.equal(dz.float().cpu().double(), dy.double() * mask)
    sc = xd64.grad.abs().max().item()
    assert (dx.float().cpu().double() - xd64.grad).abs().max().item() < tol(dt, sc) * 4
    assert torch.allclose((dgamma.cpu() - 1).double(), g64.grad, rtol=2e-3, atol=2e-3 * g64.grad.abs().max().item())
    assert torch.allclose((dbeta.cpu() - 1).double(), b64.grad, rtol=2e-3, atol=2e-3 * b64.grad.abs().max().item())


def test_bn_eval_and_f32_out():
    ops = _ops()
    D = dev()
    Cn, rows = 128, 50
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, Cn, generator=g).bfloat16()
    gamma, beta = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g)
    rm, rv = torch.randn(Cn, generator=g), torch.rand(Cn, generator=g) + 0.5
    ref = F.batch_norm(x.float(), rm, rv, gamma, beta, False, 0.1, 1e-5)
    scale, shift = torch.empty(Cn, device=D), torch.empty(Cn, device=D)
    ops.bn_eval_scale_shift(gamma.to(D), beta.to(D), rm.to(D), rv.to(D), 1e-5, Cn, scale, shift)
    y = torch.empty(rows, Cn, device=D)
    ops.bn_act(ops.dtype_code(torch.bfloat16), x.to(D), scale, shift, None, False, y, rows, Cn, out_f32=True)
    torch.cuda.synchronize()
    assert torch.allclose(y.cpu(), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_stem_im2col_matches_conv(dt):
    ops = _ops()
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 30, 26
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    ref = F.conv2d(rnd(x, dt).double(), rnd(w, dt).double(), stride=2, padding=3)
    Ho, Wo = ref.shape[2:]
    Kpad = 192
    cols = torch.empty(N * Ho * Wo, Kpad, dtype=dt, device=dev())
    ops.stem_im2col(code, x.to(dev()), cols, Kpad)
    wm = w.permute(0, 2, 3, 1).contiguous().reshape(64, 147).to(dev())
    wf = torch.empty(64, Kpad, dtype=dt, device=dev())
    ops.weight_prep(code, wm, 64, 1, 147, wf, Kpad, None)
    d = ops.fwd_desc(code, N * Ho * Wo, 1, 1, Kpad, 64, 1, 1, 0)
    y = torch.empty(N * Ho * Wo, 64, dtype=dt, device=dev())
    ops.conv_gemm(d, cols, wf, y, None, None)
    # stem weight gradient through the padded-K path: columns >= 147 must be dropped
    dy = torch.randn(N, 64, Ho, Wo, generator=g)
    ref_dw = torch.nn.grad.conv2d_weight(rnd(x, dt).double(), (64, 3, 7, 7), rnd(dy, dt).double(), stride=2, padding=3)
    d.w_row_stride = 147
    dw = torch.zeros(64 * 147 + 64, device=dev())
    ops.conv_wgrad(d, cols, nhwc(dy, dt).reshape(-1, 64), dw)
    torch.cuda.synchronize()
    got = y.float().cpu().reshape(N, Ho, Wo, 64).permute(0, 3, 1, 2)
    assert (got.double() - ref).abs().max().item() < tol(dt, ref.abs().max().item())
    assert float(dw[64 * 147:].abs().sum()) == 0.0
    got_dw = dw[:64 * 147].cpu().reshape(64, 7, 7, 3).permute(0, 3, 1, 2)
    assert (got_dw.double() - ref_dw).abs().max().item() < tol(dt, ref_dw.abs().max().item()) * 2


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
def test_pools(dt):
    ops = _ops()
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(13)
    N, Cn, H, W = 2, 64, 11, 14
    x = F.relu(torch.randn(N, Cn, H, W, generator=g))  # many exact-zero ties, as after ReLU
    xr = rnd(x, dt).double().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    Ho, Wo = ref.shape[2:]
    xd = nhwc(x, dt)
    y = torch.empty(N, Ho, Wo, Cn, dtype=dt, device=dev())
    amax = torch.empty(N * Ho * Wo * Cn, dtype=torch.uint8, device=dev())
    ops.maxpool_fwd(code, xd, y, N, H, W, Cn, amax)
    dy = rnd(torch.randn(N, Cn, Ho, Wo, generator=g), dt)
    ref.backward(dy.double())
    dx = torch.empty_like(xd)
    ops.maxpool_bwd(code, amax, nhwc(dy, dt), dx, N, H, W, Cn)
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(y).double(), ref.detach())
    assert (from_nhwc(dx).double() - xr.grad).abs().max().item() < tol(dt, 4.0)

    # average pool
    HW = H * W
    f32 = torch.empty(N, Cn, device=dev())
    ft = torch.empty(N, Cn, dtype=dt, device=dev())
    ops.avgpool_fwd(code, xd, f32, ft, N, HW, Cn)
    want = rnd(x, dt).double().mean(dim=(2, 3))
    df = rnd(torch.randn(N, Cn, generator=g), dt)
    dxa = torch.empty_like(xd)
    ops.avgpool_bwd(code, df.to(dt).to(dev()), dxa, N, HW, Cn)
    torch.cuda.synchronize()
    assert torch.allclose(f32.cpu().double(), want, atol=1e-5)
    assert (ft.float().cpu().double() - want).abs().max().item() < tol(dt, 1.0)
    want_dx = (df.double() / HW)[:, :, None, None].expand(N, Cn, H, W)
    assert (from_nhwc(dxa).double() - want_dx).abs().max().item() < tol(dt, 1.0)


@pytest.mark.parametrize("R", [8, 64, 130])
def test_ntxent_logits_and_backward(R):
    ops = _ops()
    from oracle import sm3_oracle as O
    D = dev()
    g = torch.Generator().manual_seed(R)
    z = torch.randn(R, 128, generator=g)
    z64 = z.double().requires_grad_(True)
    ref_logits, _ = O.ntxent_logits(z64, 0.1)
    zn, inv = torch.empty(R, 128, device=D), torch.empty(R, device=D)
    logits = torch.full((R, R - 1), float("nan"), device=D)
    ops.ntxent_logits(z.to(D), 0.1, zn, inv, logits)
    torch.cuda.synchronize()
    assert (logits.cpu().double() - ref_logits.detach()).abs().max().item() < 2e-5
    # CE + its gradient
    loss = torch.zeros(1, device=D)
    dlog = torch.empty_like(logits)
    ops.ce_label0(logits, 0.5, loss, dlog)
    ref_loss = 0.5 * O.cross_entropy_zero_label(ref_logits)
    ref_loss.backward()
    dz = torch.empty(R, 128, device=D)
    ops.ntxent_logits_bwd(0, dlog, zn, inv, 0.1, dz)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref_loss)) < 2e-5
    sc = z64.grad.abs().max().item()
    assert (dz.cpu().double() - z64.grad).abs().max().item() < 1e-4 * sc + 1e-7

    # fused path: same loss and gradient without the logits tensor
    ws = torch.empty(ops.ntxent_workspace_floats(R, 128), device=D)
    loss2 = torch.zeros(1, device=D)
    dz2 = torch.empty(R, 128, device=D)
    ops.ntxent_fused(0, z.to(D), 0.1, 0.5, ws, loss2, dz2)
    torch.cuda.synchronize()
    assert abs(float(loss2) - float(ref_loss)) < 2e-5
    assert (dz2.cpu().double() - z64.grad).abs().max().item() < 1e-4 * sc + 1e-7


@pytest.mark.parametrize("R,Dm", [(8, 128), (64, 128), (130, 128), (512, 128), (1024, 128), (62, 64), (66, 36), (64, 160), (10, 130)],
                         ids=lambda v: str(v))
def test_ntxent_fused_tiled_and_row_kernels(R, Dm):
    """sm3_ntxent_fused: the tiled kernels (D % 4 == 0, D <= 128: 8 anchors x 64-candidate tiles, ragged last tiles and
    anchor groups) and the row-per-workgroup fallback (D = 160, 130) against the fp64 oracle; with a loss scale too."""
    ops = _ops()
    from oracle import sm3_oracle as O
    D = dev()
    g = torch.Generator().manual_seed(R * 7 + Dm)
    z = torch.randn(R, Dm, generator=g)
    z64 = z.double().requires_grad_(True)
    ref_logits, _ = O.ntxent_logits(z64, 0.1)
    ref_loss = 0.5 * O.cross_entropy_zero_label(ref_logits)
    ref_loss.backward()
    sc = z64.grad.abs().max().item()
    ws = torch.empty(ops.ntxent_workspace_floats(R, Dm), device=D)
    for scale in (None, 1024.0):
        loss = torch.zeros(1, device=D)
        dz = torch.full((R, Dm), float("nan"), device=D)
        ops.ntxent_fused(0, z.to(D), 0.1, 0.5, ws, loss, dz,
                         dz_scale=None if scale is None else torch.tensor([scale], device=D))
        torch.cuda.synchronize()
        assert abs(float(loss) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
        k = scale or 1.0
        assert (dz.cpu().double() / k - z64.grad).abs().max().item() < 1e-4 * sc + 1e-7
    # The loss is a fixed-order sum of per-row terms (no float atomics): repeated calls -- alone, and beside a second stream
    # that keeps the chip busy so that workgroups arrive in another order -- give the same bits, and four calls into one
    # accumulator (the four terms of a step, tools/backbone_train.py:101-102,119-121) do too.
    zd = z.to(D)
    seen = set()
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=D)
    for rep in range(12):
        loss = torch.zeros(1, device=D)
        dz = torch.empty(R, Dm, device=D)
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    big = big * 1.0001
        for w in (1.0, 1.0, 0.5, 0.5):
            ops.ntxent_fused(0, zd, 0.1, w, ws, loss, dz)
        torch.cuda.synchronize()
        seen.add(loss.cpu().numpy().tobytes())
    assert len(seen) == 1


def test_adamw_matches_torch():
    ops = _ops()
    from oracle import sm3_oracle as O
    D = dev()
    g = torch.Generator().manual_seed(17)
    n = 100003
    npad = (n + 3) // 4 * 4
    p = torch.randn(npad, generator=g)
    grads = [torch.randn(npad, generator=g) * 0.01 for _ in range(3)]
    pr, mr, vr = p.double().clone(), torch.zeros(npad, dtype=torch.float64), torch.zeros(npad, dtype=torch.float64)
    pd, md, vd = p.clone().to(D), torch.zeros(npad, device=D), torch.zeros(npad, device=D)
    for step, gr in enumerate(grads, start=1):
        O.adamw_step(pr, gr.double(), mr, vr, step, 1e-3, eps=1e-5, weight_decay=5e-2)
        ops.adamw(pd, gr.to(D), md, vd, 1e-3, 0.9, 0.999, 1e-5, 5e-2, step)
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu().double(), pr, atol=2e-6, rtol=1e-5)
    # skip-on-overflow
    found = torch.zeros(1, dtype=torch.int32, device=D)
    bad = grads[0].clone(); bad[5] = float("inf")
    ops.check_finite(bad.to(D), found)
    before = pd.clone()
    ops.adamw(pd, bad.to(D), md, vd, 1e-3, 0.9, 0.999, 1e-5, 5e-2, 4, 1.0, found)
    torch.cuda.synchronize()
    assert int(found) == 1 and torch.equal(before, pd)


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
def test_bn_two_views_in_one_launch_equal_two_launches(dt):
    """`views=2` (two row ranges back to back, per-view statistics / parameters) must give bit for bit what two
    `views=1` calls give, running statistics updated view 0 first."""
    ops = _ops()
    code = ops.dtype_code(dt)
    D = dev()
    E = 4 if dt == torch.float32 else 8
    rows, Cn, nblk = 384, 256, 3
    g = torch.Generator().manual_seed(21)
    x = torch.cat([torch.randn(rows, Cn, generator=g) * 1.5 + 0.3, torch.randn(rows, Cn, generator=g) * 0.7 - 1.0])
    xd = x.to(dt).to(D)
    res = torch.randn(2 * rows, Cn, generator=g).to(dt).to(D)
    partials = torch.zeros(2 * nblk, 2, Cn)
    xr = xd.float().cpu().double()
    for b in range(2 * nblk):
        blk = xr[b * 128:(b + 1) * 128]
        partials[b, 0], partials[b, 1] = blk.sum(0).float(), (blk * blk).sum(0).float()
    partials = partials.to(D)
    gamma, beta = (torch.rand(Cn, generator=g) + 0.5).to(D), torch.randn(Cn, generator=g).to(D)
    out = {}
    for mode in ("two_launches", "one_launch"):
        rm, rv = torch.zeros(Cn, device=D), torch.ones(Cn, device=D)
        nbt = torch.zeros((), dtype=torch.int64, device=D)
        scale, shift = torch.empty(2, Cn, device=D), torch.empty(2, Cn, device=D)
        mean, invstd = torch.empty(2, Cn, device=D), torch.empty(2, Cn, device=D)
        y = torch.empty(2 * rows, Cn, dtype=dt, device=D)
        mask = torch.empty(2 * rows * Cn // E, dtype=torch.uint8, device=D)
        if mode == "one_launch":
            ws, groups = ops.bn_stats_reduce(partials, nblk, Cn, None, views=2)
            ops.bn_finalize(ws, rows, Cn, gamma, beta, 1e-5, 0.1, rm, rv, nbt, scale, shift, mean, invstd, groups=groups, views=2)
            ops.bn_act(code, xd, scale, shift, res, True, y, rows, Cn, mask=mask, views=2)
        else:
            for v in range(2):
                ws, groups = ops.bn_stats_reduce(partials[v * nblk:(v + 1) * nblk], nblk, Cn, None)
                ops.bn_finalize(ws, rows, Cn, gamma, beta, 1e-5, 0.1, rm, rv, nbt, scale[v], shift[v], mean[v], invstd[v], groups=groups)
                sl, ms = slice(v * rows, (v + 1) * rows), slice(v * rows * Cn // E, (v + 1) * rows * Cn // E)
                ops.bn_act(code, xd[sl], scale[v], shift[v], res[sl], True, y[sl], rows, Cn, mask=mask[ms])
        # backward
        dy = torch.randn(2 * rows, Cn, generator=torch.Generator().manual_seed(5)).to(dt).to(D)
        prow = ops.bn_bwd_partial_rows(rows, Cn)
        bpart = torch.empty(2 * prow, 2, Cn, device=D)
        lsums = torch.empty(2, 2 * Cn, dtype=torch.float64, device=D)
        dx = torch.empty(2 * rows, Cn, dtype=dt, device=D)
        dgamma, dbeta = torch.zeros(Cn, device=D), torch.zeros(Cn, device=D)
        if mode == "one_launch":
            ops.bn_bwd_reduce(code, dy, None, xd, mean, invstd, dy, rows, Cn, bpart, mask=mask, views=2)
            ops.bn_stats_reduce(bpart, prow, Cn, lsums, views=2)
            ops.bn_bwd_apply(code, dy, xd, mean, invstd, gamma, lsums, rows, lsums, dgamma, dbeta, dx, rows, Cn, views=2)
        else:
            for v in range(2):
                sl, ms = slice(v * rows, (v + 1) * rows), slice(v * rows * Cn // E, (v + 1) * rows * Cn // E)
                bp = bpart[v * prow:(v + 1) * prow]
                ops.bn_bwd_reduce(code, dy[sl], None, xd[sl], mean[v], invstd[v], dy[sl], rows, Cn, bp, mask=mask[ms])
                ops.bn_stats_reduce(bp, prow, Cn, lsums[v])
                ops.bn_bwd_apply(code, dy[sl], xd[sl], mean[v], invstd[v], gamma, lsums[v], rows, lsums[v], dgamma, dbeta, dx[sl], rows, Cn)
        torch.cuda.synchronize()
        out[mode] = [t.clone() for t in (scale, shift, mean, invstd, rm, rv, nbt, y, mask, dy, lsums, dx)]
        out[mode + "_dg"] = (dgamma.clone(), dbeta.clone())
    for a, b in zip(out["one_launch"], out["two_launches"]):
        assert torch.equal(a, b)
    assert int(out["one_launch"][6]) == 2
    for a, b in zip(out["one_launch_dg"], out["two_launches_dg"]):  # float atomics of two views: order may differ
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4)
    assert not torch.equal(out["one_launch"][0][0], out["one_launch"][0][1])  # the views really differ


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("case", [(16, 256, 128, 8, 8, 1, 1, 0), (8, 64, 64, 16, 16, 3, 1, 1), (16, 128, 128, 8, 8, 3, 2, 1)])
def test_fused_dgrad_two_views_equal_two_launches(case, dt):
    """sm3_conv_dgrad_bnfuse over a batch that holds two views (per-view BN mean/invstd, per-view partial rows)
    against one launch per view."""
    ops = _ops()
    N, Ci, Co, H, W, k, s, p = case   # forward conv Ci -> Co on N images = 2 views of N/2
    code = ops.dtype_code(dt)
    E = 4 if dt == torch.float32 else 8
    g = torch.Generator().manual_seed(sum(case) + 3)
    D = dev()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, Ho, Wo, Co, generator=g).to(dt).to(D)
    w_dg = (torch.randn(Ci, k * k, Co, generator=g) / math.sqrt(Co * k * k)).to(dt).to(D)
    addend = torch.randn(N, H, W, Ci, generator=g).to(dt).to(D)
    rows = N * H * W
    bn_x = torch.randn(rows, Ci, generator=g).to(dt).to(D)
    mask = torch.randint(0, 256, (rows * Ci // E,), generator=g, dtype=torch.uint8).to(D)
    if E == 4:
        mask &= 0x0F
    mean, invstd = torch.randn(2, Ci, generator=g).to(D), (torch.rand(2, Ci, generator=g) + 0.5).to(D)
    # one launch per class over both views
    descs, full = ops.dgrad_descs(code, N, H, W, Ci, Co, k, s, p)
    assert full
    half = sum(ops.conv_partial_rows(dd) for dd in descs) // 2
    out2 = torch.empty(rows, Ci, dtype=dt, device=D)
    part2 = torch.full((2 * half, 2, Ci), float("nan"), device=D)
    off = 0
    for dd in descs:
        n = ops.conv_dgrad_bnfuse(dd, dy, w_dg, out2, addend, mask, bn_x, mean, invstd, part2, off, views=2,
                                  row_offset_view1=half + off)
        off += n // 2
    assert off == half
    # reference: each view on its own
    descs1, _ = ops.dgrad_descs(code, N // 2, H, W, Ci, Co, k, s, p)
    out1 = torch.empty(rows, Ci, dtype=dt, device=D)
    part1 = torch.full((2 * half, 2, Ci), float("nan"), device=D)
    for v in range(2):
        hs = slice(v * rows // 2, (v + 1) * rows // 2)
        off = v * half
        for dd in descs1:
            off += ops.conv_dgrad_bnfuse(dd, dy[v * N // 2:(v + 1) * N // 2], w_dg, out1[hs], addend[v * N // 2:(v + 1) * N // 2],
                                         mask[v * rows * Ci // E // 2:(v + 1) * rows * Ci // E // 2], bn_x[hs], mean[v], invstd[v],
                                         part1, off)
    torch.cuda.synchronize()
    assert torch.equal(out2, out1)
    assert torch.equal(part2, part1)


# ------------------------------------------------------------------------------------------
# round 2: BatchNorm passes folded into their consumers
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("views", [1, 2])
def test_join_with_downsample_bn_in_one_pass(dt, views):
    """sm3_bn_add_bn_act == sm3_bn_act on the downsample output followed by sm3_bn_act(+residual) on bn3's (to the
    one extra rounding the materialised identity costs), and against fp64; sm3_bn_bwd_apply2 == two sm3_bn_bwd_apply."""
    ops = _ops()
    code = ops.dtype_code(dt)
    rows, Cn = 384, 256
    E = 4 if dt == torch.float32 else 8
    g = torch.Generator().manual_seed(11 + views)
    D = dev()
    x = rnd(torch.randn(views * rows, Cn, generator=g) * 2, dt)
    x2 = rnd(torch.randn(views * rows, Cn, generator=g) + 0.3, dt)
    sc, sh = torch.rand(views, Cn, generator=g) + 0.5, torch.randn(views, Cn, generator=g) * 0.2
    sc2, sh2 = torch.rand(views, Cn, generator=g) + 0.5, torch.randn(views, Cn, generator=g) * 0.2
    ref = torch.cat([F.relu(x[v * rows:(v + 1) * rows].double() * sc[v].double() + sh[v].double()
                            + x2[v * rows:(v + 1) * rows].double() * sc2[v].double() + sh2[v].double())
                     for v in range(views)])
    xd, x2d = x.to(dt).to(D), x2.to(dt).to(D)
    y = torch.empty(views * rows, Cn, dtype=dt, device=D)
    mask = torch.empty(views * rows * Cn // E, dtype=torch.uint8, device=D)
    ops.bn_add_bn_act(code, xd, sc.to(D), sh.to(D), x2d, sc2.to(D), sh2.to(D), True, y, rows, Cn, mask=mask, views=views)
    torch.cuda.synchronize()
    assert (y.float().cpu().double() - ref).abs().max().item() < tol(dt, ref.abs().max().item())
    bits = ((mask.cpu().unsqueeze(1) >> torch.arange(E, dtype=torch.uint8)) & 1).reshape(views * rows, Cn).bool()
    assert torch.equal(bits, y.float().cpu() > 0)
    # two-pass form
    idn = torch.empty_like(y)
    ops.bn_act(code, x2d, sc2.to(D), sh2.to(D), None, False, idn, rows, Cn, views=views)
    y2 = torch.empty_like(y)
    ops.bn_act(code, xd, sc.to(D), sh.to(D), idn, True, y2, rows, Cn, views=views)
    torch.cuda.synchronize()
    assert (y.float() - y2.float()).abs().max().item() <= tol(dt, ref.abs().max().item())

    # backward: the same dz into both BatchNorms
    dz = rnd(torch.randn(views * rows, Cn, generator=g), dt).to(dt).to(D)
    sides = []
    for xx in (xd, x2d):
        mean = torch.randn(views, Cn, generator=g).to(D) * 0.1
        invstd = (torch.rand(views, Cn, generator=g) + 0.5).to(D)
        gamma = (torch.rand(Cn, generator=g) + 0.5).to(D)
        gs = torch.randn(views, 2 * Cn, generator=g, dtype=torch.float64).to(D)
        ls = torch.randn(views, 2 * Cn, generator=g, dtype=torch.float64).to(D)
        sides.append(dict(x=xx, mean=mean, invstd=invstd, gamma=gamma, gsums=gs, lsums=ls))
    out = {}
    for mode in ("single", "dual"):
        dx = [torch.empty_like(dz), torch.empty_like(dz)]
        dg = [torch.zeros(Cn, device=D), torch.zeros(Cn, device=D)]
        db = [torch.zeros(Cn, device=D), torch.zeros(Cn, device=D)]
        if mode == "single":
            for i, sd in enumerate(sides):
                ops.bn_bwd_apply(code, dz, sd["x"], sd["mean"], sd["invstd"], sd["gamma"], sd["gsums"], rows, sd["lsums"],
                                 dg[i], db[i], dx[i], rows, Cn, views=views)
        else:
            a, b = (dict(sd, dgamma=dg[i], dbeta=db[i], dx=dx[i]) for i, sd in enumerate(sides))
            ops.bn_bwd_apply2(code, dz, rows, a, b, rows, Cn, views=views)
        torch.cuda.synchronize()
        out[mode] = (dx, dg, db)
    for i in range(2):
        if dt != torch.float32:
            assert torch.equal(out["single"][0][i], out["dual"][0][i])
        else:  # f32: the two kernels contract k0*(dz-k1) - (x-mu)*q into FMAs differently (1 ulp)
            assert torch.allclose(out["single"][0][i], out["dual"][0][i], rtol=2e-6, atol=2e-6)
        assert torch.allclose(out["single"][1][i], out["dual"][1][i], rtol=1e-6, atol=1e-6)
        assert torch.allclose(out["single"][2][i], out["dual"][2][i], rtol=1e-6, atol=1e-6)
    # in place over dz on the second side (how the engine calls it)
    dzc = dz.clone()
    dx0 = torch.empty_like(dz)
    a, b = (dict(sd, dgamma=None, dbeta=None, lsums=None, dx=d) for sd, d in zip(sides, (dx0, dzc)))
    ops.bn_bwd_apply2(code, dzc, rows, a, b, rows, Cn, views=views)
    torch.cuda.synchronize()
    assert torch.equal(dx0, out["dual"][0][0]) and torch.equal(dzc, out["dual"][0][1])


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("geom", [(2, 13, 11, 1), (4, 16, 16, 2), (6, 9, 14, 2)])
def test_stem_bn_relu_maxpool_fused_equals_separate_kernels(dt, geom):
    """sm3_bn_relu_maxpool_fwd == sm3_bn_act + sm3_maxpool3x3s2_fwd (values and argmax bit for bit);
    sm3_maxpool_bn_bwd == sm3_maxpool3x3s2_bwd + sm3_bn_bwd_reduce with the stored ReLU mask (dz bit for bit, sums equal)."""
    ops = _ops()
    code = ops.dtype_code(dt)
    N, H, W, views = geom
    Cn = 64
    E = 4 if dt == torch.float32 else 8
    D = dev()
    g = torch.Generator().manual_seed(N * 100 + H)
    x = rnd(torch.randn(N, H, W, Cn, generator=g) * 1.5, dt).to(dt).to(D)
    sc = (torch.randn(views, Cn, generator=g) * 0.8).to(D)  # negative scales too
    sh = (torch.randn(views, Cn, generator=g) * 0.3).to(D)
    mean = (torch.randn(views, Cn, generator=g) * 0.1).to(D)
    invstd = (torch.rand(views, Cn, generator=g) + 0.5).to(D)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    rows = N // views * H * W
    # separate kernels
    y = torch.empty(N * H * W, Cn, dtype=dt, device=D)
    mask = torch.empty(N * H * W * Cn // E, dtype=torch.uint8, device=D)
    ops.bn_act(code, x.reshape(-1, Cn), sc, sh, None, True, y, rows, Cn, mask=mask, views=views)
    p_ref = torch.empty(N * Ho * Wo, Cn, dtype=dt, device=D)
    a_ref = torch.empty(N * Ho * Wo * Cn, dtype=torch.uint8, device=D)
    ops.maxpool_fwd(code, y, p_ref, N, H, W, Cn, a_ref)
    # fused
    p = torch.empty_like(p_ref)
    a = torch.empty_like(a_ref)
    ops.bn_relu_maxpool_fwd(code, x.reshape(-1, Cn), sc, sh, p, N, H, W, Cn, a, views=views)
    torch.cuda.synchronize()
    assert torch.equal(p, p_ref) and torch.equal(a, a_ref)
    # against torch (fp64)
    ref = torch.cat([F.relu(x[v * (N // views):(v + 1) * (N // views)].double().cpu() * sc[v].double().cpu()
                            + sh[v].double().cpu()) for v in range(views)])
    ref_p = F.max_pool2d(ref.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, Cn)
    assert (p.float().cpu().double() - ref_p).abs().max().item() < tol(dt, ref_p.abs().max().item())

    # backward
    dyp = rnd(torch.randn(N * Ho * Wo, Cn, generator=g), dt).to(dt).to(D)
    dy_full = torch.empty(N * H * W, Cn, dtype=dt, device=D)
    ops.maxpool_bwd(code, a_ref, dyp, dy_full, N, H, W, Cn)
    prow_ref = ops.bn_bwd_partial_rows(rows, Cn)
    part_ref = torch.zeros(views * prow_ref * 2 * Cn, device=D)
    dz_ref = torch.empty_like(dy_full)
    ops.bn_bwd_reduce(code, dy_full, None, x.reshape(-1, Cn), mean, invstd, dz_ref, rows, Cn, part_ref, mask=mask, views=views)
    prow = ops.maxpool_bn_bwd_partial_rows(N, H, W, views)
    part = torch.full((views * prow * 2 * Cn,), float("nan"), device=D)
    dz = torch.empty_like(dy_full)
    ops.maxpool_bn_bwd(code, a, dyp, x.reshape(-1, Cn), sc, sh, mean, invstd, dz, part, N, H, W, Cn, views=views)
    torch.cuda.synchronize()
    assert torch.equal(dz, dz_ref)
    s_ref = part_ref.reshape(views, prow_ref, 2, Cn).double().sum(1).cpu()
    s_new = part.reshape(views, prow, 2, Cn).double().sum(1).cpu()
    assert torch.allclose(s_new, s_ref, rtol=1e-4, atol=1e-3 * float(s_ref.abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("geom", [(3, 30, 26, 1), (4, 64, 64, 2), (2, 45, 270, 1)])
def test_direct_stem_forward_and_fused_weight_gradient(geom, dt):
    """csrc/stem.hip (bf16 / fp16 on the 16-bit MFMA, exact f32 on v_mfma_f32_32x32x2_f32): sm3_stem_conv_fwd against F.conv2d(7x7/2/3) in fp64 on bf16-rounded operands, its BatchNorm
    partial sums against the stored output; sm3_stem_wgrad_bn against conv2d_weight of the BatchNorm input gradient
    computed in fp64 from the same (dz, xo, sums).  Odd sizes, two views, and an image wider than one 128-pixel tile."""
    ops = _ops()
    code = ops.dtype_code(dt)
    N, H, W, views = geom
    D = dev()
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w_master = w.permute(0, 2, 3, 1).contiguous().to(D)  # [64][kh][kw][c]
    w_stem = torch.empty(64, ops.STEM_KDIRECT, dtype=dt, device=D)
    ops.stem_weight_prep(code, w_master.reshape(64, 147), w_stem)
    y = torch.empty(N * Ho * Wo, 64, dtype=dt, device=D)
    prow = ops.stem_partial_rows(N, H, W)
    part = torch.full((prow, 2, 64), float("nan"), device=D)
    ops.stem_conv_fwd(code, x.to(D), w_stem, y, part)
    torch.cuda.synchronize()
    ref = F.conv2d(rnd(x, dt).double(), rnd(w, dt).double(), stride=2, padding=3)
    got = y.float().cpu().reshape(N, Ho, Wo, 64).permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() < tol(dt, ref.abs().max().item())
    yv = y.float().double()
    ps = part.double().sum(0)
    assert torch.allclose(ps[0], yv.sum(0), rtol=1e-4, atol=1e-3)
    assert torch.allclose(ps[1], (yv * yv).sum(0), rtol=1e-4, atol=1e-3)
    # per-view partial rows are contiguous (tiles are image-major)
    pv = part.reshape(views, prow // views, 2, 64).double().sum(1)
    rows_v = N // views * Ho * Wo
    for v in range(views):
        assert torch.allclose(pv[v, 0], yv[v * rows_v:(v + 1) * rows_v].sum(0), rtol=1e-4, atol=1e-3)

    # weight gradient with the BatchNorm-backward apply on the operand path
    dz = rnd(torch.randn(N * Ho * Wo, 64, generator=g), dt)
    xo = rnd(torch.randn(N * Ho * Wo, 64, generator=g) * 2 + 0.3, dt)
    mean = torch.randn(views, 64, generator=g) * 0.2
    invstd = torch.rand(views, 64, generator=g) + 0.5
    gamma = torch.rand(64, generator=g) + 0.5
    count = float(rows_v)
    gs = torch.empty(views, 128, dtype=torch.float64)
    for v in range(views):
        dzv, xov = dz[v * rows_v:(v + 1) * rows_v].double(), xo[v * rows_v:(v + 1) * rows_v].double()
        gs[v, :64] = dzv.sum(0)
        gs[v, 64:] = (dzv * (xov - mean[v].double()) * invstd[v].double()).sum(0)
    dxo = torch.empty(N * Ho * Wo, 64, dtype=torch.float64)
    for v in range(views):
        sl = slice(v * rows_v, (v + 1) * rows_v)
        xh = (xo[sl].double() - mean[v].double()) * invstd[v].double()
        dxo[sl] = gamma.double() * invstd[v].double() * (dz[sl].double() - gs[v, :64] / count - xh * gs[v, 64:] / count)
    dxo_r = rnd(dxo.float(), dt).double()  # the kernel rounds the operand to bf16 before the MFMA
    ref_dw = torch.nn.grad.conv2d_weight(rnd(x, dt).double(), (64, 3, 7, 7),
                                         dxo_r.reshape(N, Ho, Wo, 64).permute(0, 3, 1, 2), stride=2, padding=3)
    dw = torch.zeros(64, 147, device=D)
    dgam, dbet = torch.zeros(64, device=D), torch.zeros(64, device=D)
    ops.stem_wgrad_bn(code, x.to(D), dz.to(dt).to(D), xo.to(dt).to(D), mean.to(D), invstd.to(D), gamma.to(D), gs.to(D), count,
                      gs.to(D), dgam, dbet, dw, views=views)
    torch.cuda.synchronize()
    got_dw = dw.cpu().reshape(64, 7, 7, 3).permute(0, 3, 1, 2).double()
    sc = ref_dw.abs().max().item()
    assert (got_dw - ref_dw).abs().max().item() < 2 * tol(dt, sc), ((got_dw - ref_dw).abs().max().item(), sc)
    assert torch.allclose(dbet.cpu().double(), gs[:, :64].sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(dgam.cpu().double(), gs[:, 64:].sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES3, ids=IDS3)
@pytest.mark.parametrize("geom", [(4, 10, 12), (3, 9, 7)])
def test_fused_dgrad_with_compact_stride2_addend(dt, geom):
    """sm3_conv_dgrad_bnfuse with the addend given only at the even output pixels (compact tensor) == the same launch
    with that addend scattered into a dense zero tensor: dz bit for bit, partial sums bit for bit."""
    ops = _ops()
    code = ops.dtype_code(dt)
    N, H, W = geom
    Ci, Co = 128, 64  # forward conv1: Ci -> Co (1x1); its data gradient has Ci output channels
    E = 4 if dt == torch.float32 else 8
    D = dev()
    g = torch.Generator().manual_seed(N * 10 + H)
    dy = torch.randn(N, H, W, Co, generator=g).to(dt).to(D)
    w_dg = (torch.randn(Ci, 1, Co, generator=g) / math.sqrt(Co)).to(dt).to(D)
    Hs, Ws = (H + 1) // 2, (W + 1) // 2
    sp = torch.randn(N, Hs, Ws, Ci, generator=g).to(dt).to(D)
    dense = torch.zeros(N, H, W, Ci, dtype=dt, device=D)
    dense[:, ::2, ::2, :] = sp
    bn_x = torch.randn(N * H * W, Ci, generator=g).to(dt).to(D)
    mask = torch.randint(0, 256, (N * H * W * Ci // E,), generator=g, dtype=torch.uint8).to(D)
    mean = (torch.randn(Ci, generator=g) * 0.1).to(D)
    invstd = (torch.rand(Ci, generator=g) + 0.5).to(D)
    descs, full = ops.dgrad_descs(code, N, H, W, Ci, Co, 1, 1, 0)
    assert full and len(descs) == 1
    prow = ops.conv_partial_rows(descs[0])
    outs = []
    for addend, sparse in ((dense, None), (sp, (Hs, Ws))):
        out = torch.empty(N * H * W, Ci, dtype=dt, device=D)
        part = torch.full((prow, 2, Ci), float("nan"), device=D)
        ops.conv_dgrad_bnfuse(descs[0], dy, w_dg, out, addend, mask, bn_x, mean, invstd, part, 0, addend_sparse=sparse)
        torch.cuda.synchronize()
        outs.append((out, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(ValueError):
        ops.conv_dgrad_bnfuse(descs[0], dy, w_dg, outs[0][0], sp, mask, bn_x, mean, invstd, outs[0][1], 0,
                              addend_sparse=(Hs + 1, Ws))
