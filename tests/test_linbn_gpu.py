"""BatchNorm backward by linearity (csrc/linbn.hip) on the GPU: every new entry point against fp64 PyTorch of the same
arithmetic, then the whole conv3 -> bn3 backward unit (the launch sequence of SM3Engine.conv3_backward_linbn) against the
fp64 AUTOGRAD of the reference's modules (src/models/resnet.py:162-163: conv1x1 -> BatchNorm2d in train mode), and the
encoder's gradients with the linear backward on and off."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTS = [torch.bfloat16, torch.float16]
IDS = ["bf16", "f16"]


def _ops():
    from sm3hip import ops
    return ops


def dev():
    return torch.device("cuda:0")


def tol(dt):
    return {torch.bfloat16: 1.2e-2, torch.float16: 2e-3}[dt]


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("case", [(2, 1, 640, 256, 64), (2, 2, 1024, 256, 64), (1, 2, 768, 512, 128), (3, 1, 300, 128, 64)])
def test_wgrad_cat_views(case, dt):
    """P[v] = dz_v^T y_v and G[v] = y_v^T y_v from ONE launch over two views (sm3_conv_wgrad_cat), and each of them from
    a launch of its own (the form the engine uses: G in the forward pass, P in the backward pass)."""
    ops = _ops()
    N, V, HW, C, p = case
    M = N * V * HW
    g = torch.Generator().manual_seed(M + C)
    dz = torch.randn(M, C, generator=g).to(dt)
    y = torch.relu(torch.randn(M, p, generator=g)).to(dt)
    code = ops.dtype_code(dt)
    d = ops.fwd_desc(code, N * V, HW, 1, p, C, 1, 1, 0)
    P = torch.zeros(V, C, p, device=dev())
    G = torch.zeros(V, p, p, device=dev())
    yd, dzd = y.to(dev()), dz.to(dev())
    ops.conv_wgrad_cat(d, yd, dzd, P, yd, G, views=V)
    torch.cuda.synchronize()
    Mv = M // V
    for v in range(V):
        dzv, yv = dz[v * Mv:(v + 1) * Mv].double(), y[v * Mv:(v + 1) * Mv].double()
        refP, refG = dzv.t() @ yv, yv.t() @ yv
        assert (P[v].cpu().double() - refP).abs().max() < 2e-5 * refP.abs().max() + 1e-3
        assert (G[v].cpu().double() - refG).abs().max() < 2e-5 * refG.abs().max() + 1e-3
    P2 = torch.zeros(V, C, p, device=dev())
    G2 = torch.zeros(V, p, p, device=dev())
    ops.conv_wgrad_cat(d, yd, dzd, P2, views=V)
    ops.conv_wgrad_cat(ops.fwd_desc(code, N * V, HW, 1, p, p, 1, 1, 0), yd, yd, G2, views=V)
    torch.cuda.synchronize()
    assert torch.allclose(P2, P, rtol=1e-5, atol=1e-4 * P.abs().max().item())
    assert torch.allclose(G2, G, rtol=1e-5, atol=1e-4 * G.abs().max().item())
    # plain-store split-K: slabs summed in a fixed order -- no atomics, so two launches agree bit for bit, and so does a
    # launch over ONE view with that view's share of the two-view launch
    n = C * p
    slabs = torch.full((V * ops.SLAB_CAP * n,), float("nan"), device=dev())
    ns = ops.conv_wgrad_slabs(d, yd, dzd, slabs, views=V)
    assert 1 <= ns <= ops.SLAB_CAP
    P3 = torch.empty(V, C, p, device=dev())
    ops.linbn_moments(slabs, ns, n, P3, views=V)
    slabs_b = torch.full_like(slabs, float("nan"))
    assert ops.conv_wgrad_slabs(d, yd, dzd, slabs_b, views=V) == ns
    torch.cuda.synchronize()
    assert torch.equal(slabs[: V * ns * n], slabs_b[: V * ns * n])
    assert torch.allclose(P3, P, rtol=1e-5, atol=1e-4 * P.abs().max().item())
    d1 = ops.fwd_desc(code, N, HW, 1, p, C, 1, 1, 0)
    for v in range(V):
        s1 = torch.full((ops.SLAB_CAP * n,), float("nan"), device=dev())
        assert ops.conv_wgrad_slabs(d1, yd[v * Mv:(v + 1) * Mv], dzd[v * Mv:(v + 1) * Mv], s1, views=1) == ns
        Pv = torch.empty(1, C, p, device=dev())
        ops.linbn_moments(s1, ns, n, Pv, views=1)
        torch.cuda.synchronize()
        assert torch.equal(Pv[0], P3[v])


@pytest.mark.parametrize("dt", DTS + [torch.float32], ids=IDS + ["f32"])
@pytest.mark.parametrize("case", [(1, 5000, 64), (2, 1664, 128), (2, 777, 512), (1, 40, 2048)])
def test_bn_act_colsum(case, dt):
    """sm3_bn_act_colsum: same output and mask as sm3_bn_act, plus per-block column sums of the STORED values whose
    fixed-order total (sm3_linbn_moments) is bit-identical whether a view shares the launch with its sibling or not."""
    ops = _ops()
    V, rows, C = case
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(V * rows, C, generator=g).to(dt).to(dev())
    scale = (torch.rand(V * C, generator=g) + 0.5).to(dev())
    shift = (torch.randn(V * C, generator=g) * 0.3).to(dev())
    code = ops.dtype_code(dt)
    E = 4 if dt == torch.float32 else 8
    y0 = torch.empty_like(x)
    m0 = torch.empty(V * rows * C // E, dtype=torch.uint8, device=dev())
    ops.bn_act(code, x, scale, shift, None, True, y0, rows, C, mask=m0, views=V)
    crow = ops.bn_act_colsum_rows(code, rows, C)
    assert crow >= 1
    cs = torch.full((V, crow, C), float("nan"), device=dev())
    y1 = torch.empty_like(x)
    m1 = torch.empty_like(m0)
    ops.bn_act(code, x, scale, shift, None, True, y1, rows, C, mask=m1, views=V, colsum=cs)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(m0, m1)
    want = y1.double().reshape(V, rows, C).sum(1).cpu()
    assert torch.allclose(cs.double().sum(1).cpu(), want, rtol=1e-5, atol=1e-3)
    # one view at a time: the same partial rows, bit for bit
    for v in range(V):
        csv = torch.full((1, crow, C), float("nan"), device=dev())
        yv = torch.empty(rows, C, dtype=dt, device=dev())
        ops.bn_act(code, x[v * rows:(v + 1) * rows], scale[v * C:(v + 1) * C], shift[v * C:(v + 1) * C], None, True, yv, rows, C,
                   views=1, colsum=csv)
        torch.cuda.synchronize()
        assert torch.equal(csv[0], cs[v])


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("with_x", [True, False])
@pytest.mark.parametrize("case", [(1, 640, 256, 64), (2, 1024, 256, 64), (2, 512, 512, 128), (1, 333, 1024, 256),
                                  (2, 128, 2048, 512)])
def test_dgrad_two_segments_with_fused_bn_phase1(case, with_x, dt):
    """sm3_conv_dgrad_seg_bnfuse: dz_out = mask(x0 w0^T + x1 w1^T + col_bias) with per-view banks, and the producer
    BatchNorm's partial sums -- (sum dz, sum dz*xhat) with its x, (sum dz, 0) without."""
    ops = _ops()
    V, Mv, C, p = case
    M = V * Mv
    g = torch.Generator().manual_seed(M + C + p)
    x0 = torch.randn(M, C, generator=g).to(dt)
    x1 = torch.randn(M, p, generator=g).to(dt)
    w0 = (torch.randn(V, p, C, generator=g) / math.sqrt(C)).to(dt)
    w1 = (torch.randn(V, p, p, generator=g) / math.sqrt(p)).to(dt)
    bias = torch.randn(V, p, generator=g)
    bx = torch.randn(M, p, generator=g).to(dt)
    mean, invstd = torch.randn(V, p, generator=g) * 0.2, torch.rand(V, p, generator=g) + 0.5
    mask_bits = torch.rand(M, p, generator=g) > 0.4
    code = ops.dtype_code(dt)
    mb = mask_bits.reshape(M, p // 8, 8).to(torch.int32)
    mask = (mb * (1 << torch.arange(8, dtype=torch.int32))).sum(-1).to(torch.uint8).reshape(-1)
    d = ops.dgrad_descs(code, V, Mv, 1, p, C, 1, 1, 0)[0][0]
    assert (d.Ci, d.Co) == (C, p)
    prow = ops.conv_partial_rows(d)
    part = torch.full((2 * prow + 3, 2, p), float("nan"), device=dev())
    out = torch.empty(M, p, dtype=dt, device=dev())
    off1 = prow // V + 3 if V == 2 else 0
    n = ops.conv_dgrad_seg_bnfuse(d, x0.to(dev()), w0.to(dev()), x1.to(dev()), w1.to(dev()), bias.to(dev()), out,
                                  mask.to(dev()), bx.to(dev()) if with_x else None, mean.reshape(-1).to(dev()),
                                  invstd.reshape(-1).to(dev()), part, 0, views=V, row_offset_view1=off1,
                                  w_view_stride=p * C, w1_view_stride=p * p)
    torch.cuda.synchronize()
    assert n == prow
    for v in range(V):
        sl = slice(v * Mv, (v + 1) * Mv)
        ref = x0[sl].double() @ w0[v].double().t() + x1[sl].double() @ w1[v].double().t() + bias[v].double()
        ref = ref * mask_bits[sl]
        got = out[sl].double().cpu()
        sc = ref.abs().max().item()
        assert (got - ref).abs().max().item() < tol(dt) * sc
        rows_v = prow // V
        pr = part[(off1 if v else 0):(off1 if v else 0) + rows_v].double().cpu().sum(0)
        assert torch.allclose(pr[0], got.sum(0), rtol=1e-4, atol=1e-3 * sc)
        if with_x:
            xh = (bx[sl].double() - mean[v].double()) * invstd[v].double()
            want2 = (got * xh).sum(0)
            assert torch.allclose(pr[1], want2, rtol=1e-4, atol=2e-3 * sc * xh.abs().max().item())
        else:
            assert torch.equal(pr[1], torch.zeros(p, dtype=torch.float64))


def _unit_reference(y, W, gamma, beta, dz, eps=1e-5):
    """fp64 autograd of conv1x1 -> BatchNorm (train) per view; returns dy, dW, dgamma, dbeta summed over views."""
    V = y.shape[0]
    Wt = W.double().clone().requires_grad_()
    ga = gamma.double().clone().requires_grad_()
    be = beta.double().clone().requires_grad_()
    dys = []
    for v in range(V):
        yv = y[v].double().clone().requires_grad_()
        x = yv @ Wt.t()
        mu, var = x.mean(0), x.var(0, unbiased=False)
        z = (x - mu) * (var + eps).rsqrt() * ga + be
        z.backward(dz[v].double())
        dys.append(yv.grad)
    return torch.stack(dys), Wt.grad, ga.grad, be.grad


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("case", [(1, 1280, 256, 64), (2, 1024, 256, 64), (2, 640, 512, 128), (2, 256, 1024, 256),
                                  (1, 16, 2048, 512), (1, 64, 1024, 256), (1, 48, 256, 64)])
def test_conv_bn_by_linearity_matches_fp64_autograd(case, dt):
    """The launch sequences of SM3Engine.conv3_bn3_fused / conv3_backward_linbn on one conv3 -> bn3 unit against fp64
    PyTorch of conv1x1 -> BatchNorm2d(train) (-> + identity -> ReLU) (src/models/resnet.py:162-172) on the same (rounded)
    operands: batch statistics, output and ReLU bits; data gradient, weight gradient, d(gamma), d(beta)."""
    ops = _ops()
    V, Mv, C, p = case
    M = V * Mv
    g = torch.Generator().manual_seed(7 * M + C)
    code = ops.dtype_code(dt)
    y = torch.relu(torch.randn(V, Mv, p, generator=g) + 0.3).to(dt)
    Wm = torch.randn(C, p, generator=g) / math.sqrt(p)            # fp32 master [C][p]
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dz = (torch.randn(V, Mv, C, generator=g) * (torch.rand(V, Mv, C, generator=g) > 0.4)).to(dt)
    Wr = Wm.to(dt)                                                # the bank the forward used
    ref_dy, ref_dW, ref_dg, ref_db = _unit_reference(y.float(), Wr.float(), gamma, beta, dz.float())
    D = dev()
    yd, dzd = y.reshape(M, p).to(D), dz.reshape(M, C).to(D)
    w_master = Wm.to(D)
    w_fwd = torch.empty(C, p, dtype=dt, device=D)
    w_dg = torch.empty(p, 1, C, dtype=dt, device=D)
    ops.weight_prep(code, w_master, C, 1, p, w_fwd, p, w_dg)
    # sum(y) through the apply kernel (identity scale/shift reproduces y); Gram matrix through the weight-gradient kernel
    crow = ops.bn_act_colsum_rows(code, Mv, p)
    cs = torch.empty(V * crow * p, device=D)
    ytmp = torch.empty_like(yd)
    ops.bn_act(code, yd, torch.ones(V * p, device=D), torch.zeros(V * p, device=D), None, False, ytmp, Mv, p, views=V,
               colsum=cs)
    assert torch.equal(ytmp, yd)
    slabs = torch.empty(V * ops.SLAB_CAP * max(p * p, C * p), device=D)
    ns = ops.conv_wgrad_slabs(ops.fwd_desc(code, V, Mv, 1, p, p, 1, 1, 0), yd, yd, slabs, views=V)
    G = torch.empty(V * p * p, device=D)
    sd = torch.empty(V * p, dtype=torch.float64, device=D)
    ops.linbn_moments(slabs, ns, p * p, G, views=V, colsum=cs, colsum_rows=crow, s_out=sd, p=p)
    # ---- forward: bn3's batch statistics from the moments, W G saved
    Tm = torch.empty(V * C * p, device=D)
    groups = p // 32
    fws = torch.empty(V * groups * 2 * C, dtype=torch.float64, device=D)
    assert ops.linbn_fwd_stats(code, G, w_dg, w_fwd, sd, Tm, fws, C, p, V) == groups
    scale, shift = torch.empty(V * C, device=D), torch.empty(V * C, device=D)
    mean_d, invstd_d = torch.empty(V * C, device=D), torch.empty(V * C, device=D)
    rm, rv = torch.zeros(C, device=D), torch.ones(C, device=D)
    nbt = torch.zeros(1, dtype=torch.int64, device=D)
    ops.bn_finalize(fws, Mv, C, gamma.to(D), beta.to(D), 1e-5, 0.1, rm, rv, nbt, scale, shift, mean_d, invstd_d,
                    groups=groups, views=V)
    torch.cuda.synchronize()
    xe = y.double() @ Wr.double().t()                                   # un-rounded conv output [V][Mv][C]
    mu_e, var_e = xe.mean(1), xe.var(1, unbiased=False)
    assert torch.allclose(sd.reshape(V, p).cpu(), y.double().sum(1), rtol=1e-5)  # fp32 per thread, fp64 across blocks
    assert torch.allclose(mean_d.reshape(V, C).cpu().double(), mu_e, rtol=1e-4, atol=1e-5)
    assert torch.allclose(invstd_d.reshape(V, C).cpu().double(), (var_e + 1e-5).rsqrt(), rtol=2e-4)
    Gd = G.reshape(V, p, p).double().cpu()
    assert torch.allclose(Tm.reshape(V, C, p).double().cpu(), torch.einsum("ck,vkj->vcj", Wr.double(), Gd), rtol=1e-5,
                          atol=1e-5 * Gd.abs().max().item())
    assert int(nbt) == V
    # ... and conv3 -> bn3 -> +identity -> ReLU in one launch against conv -> BatchNorm2d(train) -> add -> relu in fp64
    idn = torch.randn(V, Mv, C, generator=g).to(dt)
    y3 = torch.empty(M, C, dtype=dt, device=D)
    mask3 = torch.empty(M * C // 8, dtype=torch.uint8, device=D)
    ops.conv_bn_act_fused(ops.fwd_desc(code, V, Mv, 1, p, C, 1, 1, 0), yd, w_fwd, scale, shift, idn.reshape(M, C).to(D),
                          True, y3, mask3, views=V)
    torch.cuda.synchronize()
    z = (xe - mu_e[:, None]) * (var_e[:, None] + 1e-5).rsqrt() * gamma.double() + beta.double() + idn.double()
    ref3 = torch.relu(z)
    got3 = y3.reshape(V, Mv, C).double().cpu()
    assert (got3 - ref3).abs().max().item() < tol(dt) * ref3.abs().max().item()
    bits = (mask3.cpu().reshape(M, C // 8, 1) >> torch.arange(8, dtype=torch.uint8)) & 1
    assert torch.equal(bits.reshape(V, Mv, C).bool(), y3.reshape(V, Mv, C).cpu() > 0)

    # ---- backward.  sum(dz) partial rows as a producing data-gradient epilogue leaves them (mask + sum only: x = None)
    prow = ops.bn_bwd_partial_rows(Mv, C)
    bpart = torch.empty(V * prow * 2 * C, device=D)
    ops.bn_bwd_reduce(code, dzd, None, None, None, None, None, Mv, C, bpart, views=V)
    xd = xe.reshape(M, C).to(dt).to(D)                                  # the tensor the two-pass form would have stored
    bpart2 = torch.empty_like(bpart)
    ops.bn_bwd_reduce(code, dzd, None, xd, mean_d, invstd_d, None, Mv, C, bpart2, views=V)
    want = torch.empty(V * 2 * C, dtype=torch.float64, device=D)
    ops.bn_stats_reduce(bpart2, prow, C, want, views=V)
    want = want.reshape(V, 2, C).clone()
    lsums = torch.full((V * 2 * C,), float("nan"), dtype=torch.float64, device=D)  # the linear path writes both halves
    ns = ops.conv_wgrad_slabs(ops.fwd_desc(code, V, Mv, 1, p, C, 1, 1, 0), yd, dzd, slabs, views=V)
    P = torch.empty(V * C * p, device=D)
    ops.linbn_moments(slabs, ns, C * p, P, views=V)
    ws, rgroups = ops.bn_stats_reduce(bpart, prow, C, None, views=V)
    dgamma, dbeta = torch.zeros(C, device=D), torch.zeros(C, device=D)
    coef = torch.empty(V * 4 * C, device=D)
    ops.linbn_stats(code, P, w_fwd, mean_d, invstd_d, gamma.to(D), ws, rgroups, lsums, dgamma, dbeta, Mv, coef, C, p, V)
    torch.cuda.synchronize()
    got = lsums.reshape(V, 2, C)
    assert torch.allclose(got[:, 0], want[:, 0], rtol=1e-12, atol=0)  # stage B of the reduction, folded into linbn_stats
    assert torch.allclose(got[:, 1], want[:, 1], rtol=2e-2, atol=2e-2 * want[:, 1].abs().max().item())  # stored x vs y W^T
    coef2 = torch.empty_like(coef)                                      # the data-parallel route to the same coefficients
    ops.linbn_coef(lsums, Mv, gamma.to(D), mean_d, invstd_d, coef2, C, V)
    torch.cuda.synchronize()
    assert torch.equal(coef, coef2)

    wa = torch.empty(V * p * C, dtype=dt, device=D)
    wbn = torch.empty(V * p * C, dtype=dt, device=D)
    cconst = torch.empty(V * p, device=D)
    ops.linbn_banks(code, w_dg, coef, wa, wbn, cconst, C, p, V)
    Hn = torch.empty(V * p * p, dtype=dt, device=D)
    dW = torch.zeros(C, p, device=D)
    ops.linbn_post(code, wbn, w_dg, Hn, P, G, Tm, sd, coef, dW, C, p, V)
    Hn2 = torch.empty_like(Hn)
    dW2 = torch.zeros(C, p, device=D)
    ops.linbn_post(code, wbn, w_dg, Hn2, P, G, None, sd, coef, dW2, C, p, V)   # W G recomputed in the kernel
    # banks + post as ONE launch (sm3_linbn_banks_post, what the engine runs since round 5): the same four outputs, bit for bit
    wa3, cconst3 = torch.empty_like(wa), torch.empty_like(cconst)
    Hn3, dW3 = torch.empty_like(Hn), torch.zeros(C, p, device=D)
    ops.linbn_banks_post(code, w_dg, coef, wa3, cconst3, Hn3, P, G, Tm, sd, dW3, C, p, V)
    torch.cuda.synchronize()
    assert torch.equal(Hn, Hn2) and torch.equal(dW, dW2)
    assert torch.equal(wa3, wa) and torch.equal(cconst3, cconst) and torch.equal(Hn3, Hn) and torch.equal(dW3, dW)
    # the two small products of the post kernel against the generic gather-GEMM / fp64
    Hn_ref = torch.empty(V * p * p, dtype=dt, device=D)
    ops.conv_gemm(ops.fwd_desc(code, V * p, 1, 1, C, p, 1, 1, 0), wbn, w_dg, Hn_ref)
    torch.cuda.synchronize()
    hs = Hn_ref.float().abs().max().item()
    assert (Hn.float() - Hn_ref.float()).abs().max().item() <= tol(dt) * hs
    cf = coef.reshape(V, C, 4).double().cpu()
    Wd, Pd, sdd = Wr.double(), P.reshape(V, C, p).double().cpu(), sd.reshape(V, p).cpu()
    dW_ref = sum(cf[v, :, 0, None] * (Pd[v] - cf[v, :, 2, None] * sdd[v][None]) -
                 cf[v, :, 1, None] * (Wd @ Gd[v] - cf[v, :, 3, None] * sdd[v][None]) for v in range(V))
    assert (dW.double().cpu() - dW_ref).abs().max().item() < 2e-5 * dW_ref.abs().max().item()
    dd = ops.dgrad_descs(code, V, Mv, 1, p, C, 1, 1, 0)[0][0]
    dy = torch.empty(M, p, dtype=dt, device=D)
    part = torch.empty(ops.conv_partial_rows(dd) * 2 * p, device=D)
    ops.conv_dgrad_seg_bnfuse(dd, dzd, wa, yd, Hn, cconst, dy, None, None, None, None, part, 0, views=V,
                              row_offset_view1=ops.conv_partial_rows(dd) // V, w_view_stride=p * C, w1_view_stride=p * p)
    torch.cuda.synchronize()

    t = tol(dt)
    got_dy = dy.reshape(V, Mv, p).double().cpu()
    assert (got_dy - ref_dy).abs().max().item() < 2 * t * ref_dy.abs().max().item()
    assert ((got_dy - ref_dy).norm() / ref_dy.norm()).item() < t
    assert ((dW.double().cpu() - ref_dW).norm() / ref_dW.norm()).item() < t
    assert (dW.double().cpu() - ref_dW).abs().max().item() < 2 * t * ref_dW.abs().max().item()
    assert torch.allclose(dbeta.double().cpu(), ref_db, rtol=1e-4, atol=1e-3 * ref_db.abs().max().item())
    assert ((dgamma.double().cpu() - ref_dg).norm() / ref_dg.norm()).item() < t


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("views", [1, 2])
def test_encoder_with_and_without_batchnorm_by_linearity(dt, views):
    """A whole ResNet-50 encoder forward + backward with conv3 -> bn3 by linearity against the two-pass BatchNorm form,
    same weights and inputs, both measured against the exact-f32 mode of the same engine: features, running statistics
    and gradients of the linear form are no further from the f32 ones than those of the two-pass form (the two differ only
    in where 16-bit roundings fall -- the linear form never rounds conv3's output; through 53 train-mode BatchNorms either
    is several per cent away from f32 in the earliest layers' gradients)."""
    from src.models import resnet
    from sm3hip.engine import SM3Engine
    torch.manual_seed(3)
    N = 8 if views == 1 else 64  # two views in one pass: every map of a view a multiple of 128 rows (2 x 2 at the end)
    x = torch.randn(N, 3, 64, 64).to(dev())
    res = {}
    for mode, lin in (("f32", False), ("off", False), ("on", True)):
        torch.manual_seed(5)
        m = resnet.resnet50()
        m.fc = torch.nn.Identity()
        m.to(dev()).train()
        eng = SM3Engine(m, dtype=torch.float32 if mode == "f32" else dt, kind="encoder")
        eng.linbn = lin
        eng.prepare(dev())
        eng.refresh_weights()
        plan = eng.branches["main"][0]
        f32 = torch.empty(N, 2048, device=dev())
        ctx = []
        if views == 2 and not eng.pair_ok(N // 2, 64, 64):
            pytest.skip("batch too small for two views in one pass")
        eng.encoder_forward(plan, x, True, f32, None, ctx, views=views)
        gen = torch.Generator().manual_seed(11)
        dfeat = (torch.randn(N, 2048, generator=gen) * 1e-2).to(eng.tdt).to(dev())
        eng.store.flat_g.zero_()
        eng.encoder_backward(ctx[0], dfeat)
        torch.cuda.synchronize()
        res[mode] = (f32.clone(), eng.store.flat_g.clone(),
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}, eng.store)
    fref = res["f32"][0].double()
    f_off = ((res["off"][0].double() - fref).norm() / fref.norm()).item()
    f_on = ((res["on"][0].double() - fref).norm() / fref.norm()).item()
    print(f"features, relative error vs f32: two-pass {f_off:.5f}, linear {f_on:.5f}")
    assert f_on < 1.25 * f_off + 1e-4
    for k, ref_b in res["f32"][2].items():
        if "num_batches" in k:
            assert torch.equal(res["on"][2][k], ref_b), k
            continue
        sc = ref_b.double().abs().max().item() + 1e-12
        a = (res["off"][2][k].double() - ref_b.double()).abs().max().item() / sc
        b = (res["on"][2][k].double() - ref_b.double()).abs().max().item() / sc
        assert b < 1.5 * a + 2e-3, (k, a, b)
    st = res["on"][3]
    ref = res["f32"][1].double()
    e_off = ((res["off"][1].double() - ref).norm() / ref.norm()).item()
    e_on = ((res["on"][1].double() - ref).norm() / ref.norm()).item()
    print(f"whole-gradient relative error vs f32: two-pass {e_off:.4f}, linear {e_on:.4f}")
    assert e_on < 1.25 * e_off + 1e-3
    worse = []
    for name in st.names:
        r = st._view(ref, name)
        a = ((st._view(res["off"][1], name).double() - r).norm() / (r.norm() + 1e-30)).item()
        b = ((st._view(res["on"][1], name).double() - r).norm() / (r.norm() + 1e-30)).item()
        if b > 2.0 * a + 2e-2:
            worse.append((name, a, b))
    assert not worse, worse[:8]


@pytest.mark.parametrize("dt", DTS + [torch.float32], ids=IDS + ["f32"])
@pytest.mark.parametrize("case", [(2, 4, 14, 14, 256, 2), (1, 6, 9, 7, 64, 2), (2, 2, 8, 8, 512, 1), (1, 3, 5, 11, 128, 1)])
def test_subsample_colsum(case, dt):
    """sm3_subsample_colsum: the stride-th pixels of an NHWC map as a compact tensor + the column sums of exactly those rows
    (partial rows, view-independent cut)."""
    ops = _ops()
    V, Nv, H, W, C, stride = case
    N = V * Nv
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, H, W, C, generator=g).to(dt)
    code = ops.dtype_code(dt)
    Hs, Ws = (H - 1) // stride + 1, (W - 1) // stride + 1
    want = x[:, ::stride, ::stride, :].contiguous()
    rows = Nv * Hs * Ws
    crow = ops.subsample_colsum_rows(code, rows, C)
    cs = torch.full((V, crow, C), float("nan"), device=dev())
    y = torch.empty(N, Hs, Ws, C, dtype=dt, device=dev()) if stride > 1 else None
    ops.subsample_colsum(code, x.to(dev()), y, cs, N, H, W, C, stride, views=V)
    torch.cuda.synchronize()
    if y is not None:
        assert torch.equal(y.cpu(), want)
    ref = want.double().reshape(V, rows, C).sum(1)
    assert torch.allclose(cs.double().sum(1).cpu(), ref, rtol=1e-5, atol=1e-3)
    s = torch.empty(V, C, dtype=torch.float64, device=dev())
    out = torch.empty(V, 4, device=dev())
    ops.linbn_moments(torch.zeros(V * 4, device=dev()), 1, 4, out, views=V, colsum=cs, colsum_rows=crow, s_out=s, p=C)
    torch.cuda.synchronize()
    assert torch.allclose(s.cpu(), ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("addend", [False, True])
@pytest.mark.parametrize("case", [(1, 640, 256, 64), (2, 512, 512, 256), (2, 128, 2048, 1024)])
def test_two_segment_gemm_without_a_fused_batchnorm(case, addend, dt):
    """sm3_conv_gather_gemm_seg: y = x0 w0^T + x1 w1^T + col_bias (+ addend, in place) with per-view banks -- the data
    gradient of the downsample conv -> BatchNorm pair."""
    ops = _ops()
    V, Mv, C, p = case
    M = V * Mv
    g = torch.Generator().manual_seed(M + C + p + 1)
    x0 = torch.randn(M, C, generator=g).to(dt)
    x1 = torch.randn(M, p, generator=g).to(dt)
    w0 = (torch.randn(V, p, C, generator=g) / math.sqrt(C)).to(dt)
    w1 = (torch.randn(V, p, p, generator=g) / math.sqrt(p)).to(dt)
    bias = torch.randn(V, p, generator=g)
    add = torch.randn(M, p, generator=g).to(dt)
    code = ops.dtype_code(dt)
    d = ops.fwd_desc(code, V, Mv, 1, C, p, 1, 1, 0)
    y = add.clone().to(dev()) if addend else torch.empty(M, p, dtype=dt, device=dev())
    ops.conv_gemm_seg(d, x0.to(dev()), w0.to(dev()), x1.to(dev()), w1.to(dev()), bias.to(dev()), y, y if addend else None,
                      views=V, w_view_stride=p * C, w1_view_stride=p * p)
    torch.cuda.synchronize()
    for v in range(V):
        sl = slice(v * Mv, (v + 1) * Mv)
        ref = x0[sl].double() @ w0[v].double().t() + x1[sl].double() @ w1[v].double().t() + bias[v].double()
        if addend:
            ref = ref + add[sl].double()
        sc = ref.abs().max().item()
        assert (y[sl].double().cpu() - ref).abs().max().item() < 1.5 * tol(dt) * sc


@pytest.mark.parametrize("dt", DTS, ids=IDS)
@pytest.mark.parametrize("case", [(1, 640, 256, 64, 128), (2, 512, 512, 128, 256), (2, 128, 2048, 512, 1024),
                                  (1, 16, 2048, 512, 1024), (1, 64, 1024, 256, 512)])
def test_join_of_a_downsample_block_by_linearity(case, dt):
    """The forward join of a Bottleneck with a downsample branch (resnet.py:162-172) as the engine's join_fused runs it:
    batch statistics of bn3 and of the downsample BatchNorm from the moments of their convolutions' inputs
    (sm3_linbn_fwd_stats), scales folded into the banks (sm3_linbn_scale_banks), one two-segment GEMM with ReLU + mask
    (sm3_conv_seg_act) -- against fp64 train-mode BatchNorm of the two explicit convolutions, per view."""
    ops = _ops()
    V, Mv, C, p, Cin = case
    M = V * Mv
    g = torch.Generator().manual_seed(M + C + p + Cin)
    y2 = torch.relu(torch.randn(M, p, generator=g) + 0.3).to(dt)
    xin = torch.relu(torch.randn(M, Cin, generator=g)).to(dt)
    w3 = (torch.randn(C, p, generator=g) / math.sqrt(p)).to(dt)
    wd = (torch.randn(C, Cin, generator=g) / math.sqrt(Cin)).to(dt)
    gam3, bet3 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    gamd, betd = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    code = ops.dtype_code(dt)
    D = dev()
    scales, shifts, means = [], [], []
    for x, w, K, gam, bet in ((y2, w3, p, gam3, bet3), (xin, wd, Cin, gamd, betd)):
        xd = x.to(D)
        slabs = torch.empty(64 * V * K * K, device=D)
        ns = ops.conv_wgrad_slabs(ops.fwd_desc(code, V, Mv, 1, K, K, 1, 1, 0), xd, xd, slabs, views=V)
        G = torch.empty(V * K * K, device=D)
        crow = ops.subsample_colsum_rows(code, Mv, K)
        cs = torch.empty(V * crow * K, device=D)
        ops.subsample_colsum(code, xd, None, cs, V, Mv, 1, K, 1, V)
        s = torch.empty(V * K, dtype=torch.float64, device=D)
        ops.linbn_moments(slabs, ns, K * K, G, views=V, colsum=cs, colsum_rows=crow, s_out=s, p=K)
        Tm = torch.empty(V * C * K, device=D)
        groups = K // 32
        ws = torch.empty(V * groups * 2 * C, dtype=torch.float64, device=D)
        ops.linbn_fwd_stats(code, G, w.t().contiguous().to(D), w.to(D), s, Tm, ws, C, K, V)
        sc, sh = torch.empty(V * C, device=D), torch.empty(V * C, device=D)
        mean, invstd = torch.empty(V * C, device=D), torch.empty(V * C, device=D)
        ops.bn_finalize(ws, Mv, C, gam.to(D), bet.to(D), 1e-5, 0.1, None, None, None, sc, sh, mean, invstd, groups=groups,
                        views=V)
        scales.append(sc); shifts.append(sh); means.append(mean)
    w3s = torch.empty(V * C * p, dtype=dt, device=D)
    wds = torch.empty(V * C * Cin, dtype=dt, device=D)
    bias = torch.empty(V * C, device=D)
    ops.linbn_scale_banks(code, w3.to(D), scales[0], shifts[0], w3s, wd.to(D), scales[1], shifts[1], wds, bias, C, V)
    out = torch.empty(M, C, dtype=dt, device=D)
    mask = torch.empty(M * C // 8, dtype=torch.uint8, device=D)
    ops.conv_seg_act(ops.fwd_desc(code, V, Mv, 1, p, C, 1, 1, 0), y2.to(D), w3s, xin.to(D), wds, bias, out, mask, True,
                     views=V, w_view_stride=C * p, w1_view_stride=C * Cin)
    torch.cuda.synchronize()
    t = tol(dt)
    for v in range(V):
        sl = slice(v * Mv, (v + 1) * Mv)
        ref = 0
        for x, w, gam, bet, mean in ((y2, w3, gam3, bet3, means[0]), (xin, wd, gamd, betd, means[1])):
            z = x[sl].double() @ w.double().t()
            mu, var = z.mean(0), z.var(0, unbiased=False)
            assert torch.allclose(mean.view(V, C)[v].double().cpu(), mu, rtol=1e-4, atol=1e-4 * z.abs().max().item())
            ref = ref + (z - mu) / torch.sqrt(var + 1e-5) * gam.double() + bet.double()
        ref = torch.relu(ref)
        got = out[sl].double().cpu()
        assert (got - ref).abs().max().item() < 3 * t * ref.abs().max().item()
        bits = ((mask.cpu().reshape(M, C // 8, 1) >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(M, C)[sl].bool()
        assert torch.equal(bits, out[sl].cpu() > 0)
