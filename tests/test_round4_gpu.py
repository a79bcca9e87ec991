"""Round 4: BASELINE.json configs[3] -- tools/mlc_train.py at its own per-GPU size (batch 512 over 8 ranks = 64
per GPU, 224 x 224, run.sh:39-47 settings) -- exercised on the one-GPU test box: the step of mlc_train.py:236-262 against an
fp64 restatement of the heads on the same extractor features, and the same tool with TWO ranks through cluster_memory's
gather -> rank-0 k-means -> broadcast (mlc_train.py:116-189)."""
import importlib.util
import math
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
RUN_SH = ["--mlc-proj", "v4", "--mlc-proj-dim", "512", "--num-heads", "1", "--sa-dim-ff", "128", "--sa-dropout", "0.1",
          "--temperature", "1", "-lr", "1e-4", "--num-labels", "8", "--extractor-proj-dim", "128"]  # run.sh:39-47


def _tool():
    spec = importlib.util.spec_from_file_location("sm3_mlc_train", os.path.join(ROOT, "skin-sm3_amd", "tools", "mlc_train.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    return mt


def _heads_fp64(P, feats, nhead=1):
    """mlc_train.py:75-90 on given extractor features, fp64: 8 Linear label projectors -> one TransformerEncoderLayer over the
    8 tokens (dropout inactive) -> 8 bias-free prototype heads."""
    from oracle import sm3_oracle as O
    tokens = torch.stack([F.linear(feats, P[f"projectors.projectors.{i}.0.weight"], P[f"projectors.projectors.{i}.0.bias"])
                          for i in range(8)], dim=0)
    sa = O.transformer_encoder_layer(tokens, P, "mlc_sa.", nhead)
    return sa, [F.linear(sa[i % sa.shape[0]], P[f"prototypes.{i}.weight"]) for i in range(8)]


def test_config4_mlc_train_step_b64_224_bf16_at_size():
    """One rank's share of config 4: 64 pairs of 224 x 224 through the frozen bf16 encoders (eval mode, fused conv + BN
    epilogues) and the training heads.  (a) with the attention layer's dropout inactive the head outputs, the loss and every
    head gradient equal fp64 autograd of the restated heads on the SAME extractor features; (b) the step as run.sh runs it
    (dropout 0.1 active, AdamW lr 1e-4): finite loss, every head tensor moves, the extractor does not."""
    mt = _tool()
    from src.models.projector import MultiLabelProjector4
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(3407)
    B, S = 64, 224
    margs = mt.get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-", "-b", str(B)] + RUN_SH)
    ex = SimCLRSkinV32(arch="resnet50", proj_dim=128)
    ex.derm_backbone.projector = ex.clinic_backbone.projector = ex.cross_proj = None   # mlc_train.py:344-346
    ex.sm3_dtype = torch.bfloat16
    for p in ex.parameters():
        p.requires_grad = False
    m = mt.Model(ex, MultiLabelProjector4(4096, margs.mlc_proj_dim, 8), margs.mlc_proj_dim, False, margs.num_heads,
                 margs.sa_dim_ff, margs.sa_dropout).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(7)
    derm, clinic = mt.synthetic_split(B, (S, S), torch.device(DEV), 11)
    assign = [torch.randint(0, n, (B,), device=DEV, generator=g) for n in mt.NUM_CLASSES]
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100)
    before = {k: v.clone() for k, v in m.extractor.state_dict().items()}

    # (a) parity, dropout inactive
    m.eval()
    with torch.no_grad():
        feats = torch.cat(m.extractor.extract(derm, clinic), dim=1)
    assert feats.shape == (B, 4096) and feats.dtype == torch.float32 and bool(torch.isfinite(feats).all())
    sa, preds = m(derm, clinic)
    loss = sum(crit(p / margs.temperature, a) for p, a in zip(preds, assign)) / 8
    loss.backward()
    torch.cuda.synchronize()
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()
         if k.startswith(("projectors.", "mlc_sa.", "prototypes.")) and v.is_floating_point()}
    sa_ref, preds_ref = _heads_fp64(P, feats.cpu().double())
    loss_ref = sum(F.cross_entropy(p / margs.temperature, a.cpu()) for p, a in zip(preds_ref, assign)) / 8
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref))), (float(loss), float(loss_ref))
    assert float((sa.detach().cpu().double() - sa_ref.detach()).abs().max()) < 1e-4 * max(1.0, float(sa_ref.abs().max()))
    for p, pr in zip(preds, preds_ref):
        assert float((p.detach().cpu().double() - pr.detach()).abs().max()) < 1e-4 * max(1.0, float(pr.abs().max()))
    worst = 0.0
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        gref = P[k].grad
        err = float((p.grad.cpu().double() - gref).norm() / (gref.norm() + 1e-12))
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    print(f"config 4 at size: loss {float(loss):.6f} (fp64 {float(loss_ref):.6f}), worst head-gradient rel. error {worst:.2e}")

    # (b) the step as the tool runs it (mlc_train.py:230-262)
    m.extractor.eval(); m.projectors.train(); m.mlc_sa.train(); m.prototypes.train()
    heads0 = {k: p.detach().clone() for k, p in m.named_parameters() if p.requires_grad}
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=margs.base_lr, weight_decay=margs.wd)
    _, preds = m(derm, clinic)
    loss = sum(crit(p / margs.temperature, a) for p, a in zip(preds, assign)) / 8
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and 0.0 < float(loss) < 20.0
    assert all(not torch.equal(p.detach(), heads0[k]) for k, p in m.named_parameters() if p.requires_grad)
    after = m.extractor.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before), "the frozen extractor (parameters AND BatchNorm buffers) moved"


def _rank_main(rank, world, port, q, log_path):
    try:
        os.environ["SM3_DIST_BACKEND"] = "gloo"   # two ranks on ONE GPU: RCCL refuses a device twice
        os.environ["SM3_FORCE_DEVICE"] = "0"
        for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(4)
        mt = _tool()
        args = mt.get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-", "--epochs", "1", "-b", "128", "--num-samples", "128",
                                           "--img-sz", "224", "224", "--amp", "--log-path", log_path, "--save-freq", "1"]
                                          + RUN_SH)
        args.world_size, args.port, args.probe = world, port, {}
        from src.models.simclr import SimCLRSkinV32
        torch.manual_seed(args.seed)
        fresh = SimCLRSkinV32(arch=args.arch, proj_dim=args.extractor_proj_dim)   # what main() builds first, same seed
        ref_sd = {k: v.clone() for k, v in fresh.state_dict().items() if "projector" not in k and "cross_proj" not in k}
        hist = mt.main(rank, args)
        model = args.probe["model"]
        ext = model.extractor.state_dict()
        frozen_ok = all(torch.equal(ext[k].cpu(), v) for k, v in ref_sd.items() if k in ext)
        heads = torch.cat([p.detach().float().reshape(-1).cpu() for k, p in model.named_parameters() if p.requires_grad])
        q.put((rank, True, {"hist": hist, "assign": [a.tolist() for a in args.probe["assignments"]],
                            "protos": [float(p.double().sum()) for p in args.probe["prototypes"]],
                            "heads": [float(heads.double().sum()), float(heads.double().abs().sum())],
                            "frozen_ok": frozen_ok, "n_ext": len(ref_sd)}))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


def test_config4_two_ranks_cluster_and_step_on_one_gpu(tmp_path):
    """tools/mlc_train.py with world size 2 at 64 pairs of 224 x 224 per rank (bf16 encoders), both ranks on the test box's one
    GPU over gloo, real kernels: each rank's memory bank is gathered on rank 0, clustered there (spherical k-means on
    csrc/heads_train.hip) and broadcast -- both ranks must end with IDENTICAL pseudo-labels for all 128 samples and identical
    prototypes; the DistributedDataParallel step then leaves identical heads on both ranks, a finite loss, and the frozen
    extractor exactly as it was built."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, ok, payload = q.get(timeout=180)
        assert ok, f"rank {r} failed:\n{payload}"
        res[r] = payload
    for p in procs:
        p.join(timeout=60)
    a, b = res[0], res[1]
    assert a["assign"] == b["assign"], "pseudo-labels differ between the ranks"
    for lab, n in zip(a["assign"], [5, 3, 2, 3, 3, 3, 3, 2]):
        assert len(lab) == 128 and all(0 <= v < n for v in lab), "a sample was left without a pseudo-label (-100)"
    assert a["protos"] == b["protos"]
    assert a["heads"] == b["heads"], (a["heads"], b["heads"])     # replicas in sync after the averaged step
    for r in (a, b):
        assert len(r["hist"]) == 1 and math.isfinite(r["hist"][0]) and 0.0 < r["hist"][0] < 20.0, r["hist"]
        assert r["frozen_ok"] and r["n_ext"] > 600
    assert os.path.exists(os.path.join(str(tmp_path), "ckp_0.pth"))


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_linear_and_two_pass_batchnorm_forms_agree_at_b16(dtname):
    """the BatchNorm-by-linearity forms (csrc/linbn.hip, the 16-bit default) against the two-pass form of the SAME
    engine at B = 16 pairs of 224 x 224, where 4-image BatchNorms no longer dominate: loss, gradient norm and gradient
    direction of a whole step, each form also measured against the exact-f32 mode -- the linear form must be no further
    from f32 than the two-pass form is (+ margin), and the two forms must agree with each other at least as well as either
    agrees with f32."""
    from sm3hip.trainer import SM3Trainer
    from test_config_gpu import _build, _latent_batch
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    derm, clinic = _latent_batch(16, 224, 21)

    def grad(dtype, linbn):
        model = _build(21, dtype)
        tr = SM3Trainer(model, lr=0.0, init_scale=1024.0)
        eng = tr._engine()
        if linbn is not None:
            eng.linbn = linbn
        loss = float(tr.step(derm, clinic))
        torch.cuda.synchronize()
        g = eng.store.flat_g.double().cpu()
        if dtype == torch.float16:
            assert tr.steps_taken() == 1
            g = g / 1024.0
        del tr, model, eng
        torch.cuda.empty_cache()
        return loss, g

    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    l32, g32 = grad(torch.float32, None)
    lon, gon = grad(dt, True)
    loff, goff = grad(dt, False)
    c_on, c_off, c_between = cos(gon, g32), cos(goff, g32), cos(gon, goff)
    n32, non, noff = float(g32.norm()), float(gon.norm()), float(goff.norm())
    print(f"{dtname} B=16 224: loss f32 {l32:.4f} linear {lon:.4f} two-pass {loff:.4f}; cosine vs f32 linear {c_on:.3f} two-pass "
          f"{c_off:.3f}, between the forms {c_between:.3f}; |g| f32 {n32:.4f} linear {non:.4f} two-pass {noff:.4f}")
    # measured (r4e3): bf16 cosine vs f32 0.273 (linear) / 0.159 (two-pass), between the forms 0.300, |g| 5606 / 5322 / f32 5502,
    # losses 12.365 / 12.536 / 12.667; fp16 0.714 / 0.709, between 0.796, |g| 5354 / 5365, losses 12.756 / 12.697.  (Why a bf16
    # step keeps its norm but not its direction: tests/test_round3_gpu.py::test_16bit_modes_against_the_oracle_b16_224.)
    # round 5: with the K order of the small-grid 3x3 launches changed (chunk outer: same products, another summation order)
    # the bf16 losses read 12.181 / 12.537 -- a 0.18 move of ONE form from a re-ordering of fp32 sums.  At random init a bf16
    # loss at B = 16 is only defined to a few tenths (torch's own bf16 autocast is 0.2 - 0.85 off its fp64 run
    # here); the bound is 3x the largest difference seen between the forms (0.36), fp16's 3x its largest (0.06)
    assert abs(lon - loff) < (0.2 if dtname == "f16" else 1.0)
    assert abs(non - noff) < (0.03 if dtname == "f16" else 0.08) * noff and abs(non - n32) < 0.10 * n32
    assert c_on > c_off - 0.05 and c_on >= (0.65 if dtname == "f16" else 0.15)
    assert c_between >= min(c_on, c_off) - 0.1


def test_bench_p2p_ab_prints_a_second_record_with_its_witness():
    """`SM3_BENCH_AB_P2P=1` repeats the timed loop with the peer-to-peer SyncBN exchange and prints a
    second, line-compatible record after the contract's one.  Rehearsed on the one GPU with the data-parallel path forced
    (`SM3_BENCH_FORCE_DP=1`: one rank over the real `nccl` backend): two JSON lines, the first with the RCCL witness (ranks
    COUNTED by an all-reduce), the second with `syncbn_exchange == "p2p"` and the mailboxes' memory kind."""
    import json
    import subprocess
    env = dict(os.environ, SM3_BENCH_FORCE_DP="1", SM3_BENCH_AB_P2P="1", MASTER_PORT="29577")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "32",
                         "--img", "64", "--no-cpu-baseline", "--no-other-dtypes"], env=env, capture_output=True, text=True,
                        timeout=300)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2, pr.stdout
    first, second = lines
    w1, w2 = first["config"]["witness"], second["config"]["witness"]
    assert w1["backend"] == "nccl" and w1["rccl_ranks"] == 1 and w1["syncbn_exchange"] == "rccl" and w1["nccl_version"]
    assert w2["syncbn_exchange"] == "p2p" and w2["p2p_mailbox_memory"] in ("finegrained", "coarse") and w2["rccl_ranks"] == 1
    for rec in lines:
        assert rec["unit"] == "pairs/s" and rec["value"] > 0 and rec["n_gpus"] == 1 and rec["steps"] == 2
        assert math.isfinite(rec["config"]["loss"])
    # the same trainer keeps training on the same batch through both loops (13.4 -> 12.2 after three steps, 10.6 after six,
    # whichever exchange carries the statistics: scratch/dbg_p2p_switch.py): the second record continues the first
    assert second["config"]["loss"] < first["config"]["loss"] + 0.5


_VARIANT_ENVS = [
    # (round 6 removed the variants two rounds had measured slower -- double-buffered K loop, loader / consumer waves, 8-wave
    # tiles: scratch/r6_pruned_variants.patch -- and with them their rows of this table)
    ("default", {}),
    ("no_pointwise", {"SM3_CONV_PW": "0"}),
    ("pointwise_with_x_path", {"SM3_CONV_PW": "1"}),
    ("general_epilogue", {"SM3_CONV_LEAN": "0"}),
]


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_gather_gemm_variants_are_bit_identical(dtname, monkeypatch):
    """Every instruction-level variant of the gather-GEMM that is still built (pointwise and no-x epilogues, the general
    f32-staging epilogue) computes the same sums in the same order: on a
    3x3 and two 1x1 shapes, forward (+ BatchNorm partial sums), fused BatchNorm + identity + ReLU forward and the data
    gradient with fused BN-backward phase 1 (with and without x) give the SAME BITS under every switch -- so an A/B of two
    variants compares speed and nothing else.  (The general f32-staging epilogue, SM3_CONV_LEAN=0, rounds once instead of
    twice where an affine or an addend follows: it is compared on the plain forward only.)"""
    from sm3hip import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    code = ops.dtype_code(dt)
    D = torch.device(DEV)
    g = torch.Generator().manual_seed(77)
    cases = [(4, 256, 256, 14, 3), (2, 1024, 256, 14, 1), (3, 128, 512, 28, 1)]   # N, Ci, Co, H, k
    results = {}
    for vname, env in _VARIANT_ENVS:
        for k_ in ("SM3_CONV_PW", "SM3_CONV_LEAN"):
            monkeypatch.delenv(k_, raising=False)
        monkeypatch.setenv("SM3_CONV_HALO", "0")  # same sums in ANOTHER order: test_halo_resident_3x3_... below
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        outs = []
        for ci_, (N, Ci, Co, H, k) in enumerate(cases):
            gg = torch.Generator().manual_seed(77 + ci_)  # a case's inputs do not depend on what ran before it
            pad = k // 2
            d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, pad)
            M = N * H * H
            x = torch.randn(M, Ci, generator=gg).to(dt).to(D)
            w = (torch.randn(Co, k * k * Ci, generator=gg) / math.sqrt(k * k * Ci)).to(dt).to(D)
            y = torch.empty(M, Co, dtype=dt, device=D)
            part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=D)
            ops.conv_gemm(d, x, w, y, None, part)
            outs += [y, part]
            if vname == "general_epilogue":
                continue
            if k == 1:  # fused BatchNorm affine + identity + ReLU + ReLU bits (EPI 2)
                res = torch.randn(M, Co, generator=gg).to(dt).to(D)
                sc, sh = (torch.rand(Co, generator=gg) + 0.5).to(D), torch.randn(Co, generator=gg).to(D)
                y2 = torch.empty(M, Co, dtype=dt, device=D)
                mk = torch.zeros(M * Co // 8, dtype=torch.uint8, device=D)
                ops.conv_bn_act_fused(d, x, w, sc, sh, res, True, y2, mk, views=1)
                outs += [y2, mk]
            # data gradient with the fused phase 1: dy [M, Co] -> dz [M, Ci]
            descs, full = ops.dgrad_descs(code, N, H, H, Ci, Co, k, 1, pad)
            dy = torch.randn(M, Co, generator=gg).to(dt).to(D)
            wdg = (torch.randn(Ci, k * k * Co, generator=gg) / math.sqrt(k * k * Co)).to(dt).to(D)
            add = torch.randn(M, Ci, generator=gg).to(dt).to(D)
            bnx = torch.randn(M, Ci, generator=gg).to(dt).to(D)
            msk = torch.randint(0, 256, (M * Ci // 8,), generator=gg, dtype=torch.uint8).to(D)
            mean, istd = torch.randn(Ci, generator=gg).to(D), (torch.rand(Ci, generator=gg) + 0.5).to(D)
            for with_x in (True, False):
                dz = torch.empty(M, Ci, dtype=dt, device=D)
                total = sum(ops.conv_partial_rows(dd) for dd in descs)
                fp = torch.zeros(total * 2 * Ci, device=D)
                off = 0
                for dd in descs:
                    off += ops.conv_dgrad_bnfuse(dd, dy, wdg, dz, add, msk, bnx if with_x else None, mean, istd, fp, off)
                outs += [dz, fp]
        torch.cuda.synchronize()
        results[vname] = outs
    ref = results["default"]
    for vname, outs in results.items():
        if vname == "general_epilogue":
            for i in range(len(cases)):
                assert torch.equal(outs[2 * i], ref[_fwd_index(i, cases)]), (vname, i)
            continue
        assert len(outs) == len(ref)
        for i, (a, b) in enumerate(zip(outs, ref)):
            assert torch.equal(a, b), (vname, i, float((a.double() - b.double()).abs().max()))


@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_halo_resident_3x3_matches_the_general_gather(dtname, monkeypatch):
    """The halo-resident A image of the stride-1 3x3 launches (kVarHalo: tile rows + W + 1 pixels either side staged once per
    channel chunk, nine taps read it at a row shift, out-of-image taps read a zero row) against the general gather on the
    same inputs: forward + BatchNorm partial sums, data gradient + fused BN-backward phase 1 with and without x.  Same
    products, K-steps summed chunk-outer instead of tap-outer: one chunk (Ci = 64) must give the SAME BITS, more chunks at
    most one rounding step of the 16-bit GEMM result on a handful of elements.  Geometries: every layer width of the
    network, images that end inside a tile, M not a multiple of 128, H != W, an image smaller than the halo; the small-grid
    4-stage kernel is switched off so that the variant under test is the one that runs."""
    from sm3hip import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    code = ops.dtype_code(dt)
    D = torch.device(DEV)
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    monkeypatch.setenv("SM3_CONV_DEEP", "0")

    def run(N, H, W, Ci, Co, seed):
        gg = torch.Generator().manual_seed(seed)
        d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
        M = N * H * W
        x = torch.randn(M, Ci, generator=gg).to(dt).to(D)
        w = (torch.randn(Co, 9 * Ci, generator=gg) / math.sqrt(9 * Ci)).to(dt).to(D)
        y = torch.empty(M, Co, dtype=dt, device=D)
        part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=D)
        ops.conv_gemm(d, x, w, y, None, part)
        outs = [y, part]
        descs, _ = ops.dgrad_descs(code, N, H, W, Ci, Co, 3, 1, 1)
        dy = torch.randn(M, Co, generator=gg).to(dt).to(D)
        wdg = (torch.randn(Ci, 9 * Co, generator=gg) / math.sqrt(9 * Co)).to(dt).to(D)
        add = torch.randn(M, Ci, generator=gg).to(dt).to(D)
        bnx = torch.randn(M, Ci, generator=gg).to(dt).to(D)
        msk = torch.randint(0, 256, (M * Ci // 8,), generator=gg, dtype=torch.uint8).to(D)
        mean, istd = torch.randn(Ci, generator=gg).to(D), (torch.rand(Ci, generator=gg) + 0.5).to(D)
        for with_x in (True, False):
            dz = torch.empty(M, Ci, dtype=dt, device=D)
            fp = torch.zeros(sum(ops.conv_partial_rows(dd) for dd in descs) * 2 * Ci, device=D)
            off = 0
            for dd in descs:
                off += ops.conv_dgrad_bnfuse(dd, dy, wdg, dz, add, msk, bnx if with_x else None, mean, istd, fp, off)
            outs += [dz, fp]
        torch.cuda.synchronize()
        return outs

    cases = [(3, 14, 14, 256, 256), (2, 28, 28, 128, 128), (2, 56, 56, 64, 64), (5, 7, 7, 512, 512), (3, 10, 14, 128, 256),
             (1, 3, 5, 64, 64), (40, 14, 14, 256, 256)]
    for ci_, c in enumerate(cases):
        monkeypatch.setenv("SM3_CONV_HALO", "0")
        a = run(*c, seed=5 + ci_)
        monkeypatch.setenv("SM3_CONV_HALO", "1")
        b = run(*c, seed=5 + ci_)
        for i, (u, v) in enumerate(zip(a, b)):
            what = (c, ["y", "partials", "dz(x)", "fz partials(x)", "dz", "fz partials"][i])
            if c[3] == 64:
                assert torch.equal(u, v), what
                continue
            u, v = u.double(), v.double()
            if i % 2 == 0:  # stored 16-bit tensors: one rounding step of the larger magnitude, on few elements
                # (of the GEMM result, that is: the data gradient adds its addend AFTER that rounding, so the step is measured
                # on the tensor's scale, not on the element's own)
                assert float((u - v).abs().max()) <= 2 * ulp * float(u.abs().max()), (what, float((u - v).abs().max()))
                # (f16 rounds 8x finer than bf16, so the same f32 ordering noise crosses a rounding boundary 8x as often)
                assert float((u != v).double().mean()) < (2e-3 if dt == torch.bfloat16 else 2e-2), (what, float((u != v).double().mean()))
            else:           # f32 sums of those tensors over 128 rows
                assert float((u - v).abs().max()) <= 2e-3 * (float(u.abs().max()) + 1e-6), what
    # the variant must be the one the step uses: a forward 3x3 of the network at size, both forms against fp64
    N, H, Ci = 8, 14, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Ci, H, H, generator=g).to(dt)
    w = (torch.randn(Ci, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(dt)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Ci)
    d = ops.fwd_desc(code, N, H, H, Ci, Ci, 3, 1, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().reshape(-1, Ci).to(D)
    wd = w.permute(0, 2, 3, 1).contiguous().reshape(Ci, -1).to(D)
    y = torch.empty(N * H * H, Ci, dtype=dt, device=D)
    ops.conv_gemm(d, xd, wd, y, None, None)
    torch.cuda.synchronize()
    assert float((y.cpu().double() - ref).abs().max()) < 4 * ulp * float(ref.abs().max())


@pytest.mark.parametrize("dtname", ["bf16", "f16", "f32"])
def test_batched_weight_prep_equals_the_single_bank_kernel(dtname):
    """sm3_weight_prep_batch (all filter banks of a lane in one launch; 16-byte reads / packed writes and a 64 x 32 paired
    transpose for the 16-bit banks) against sm3_weight_prep bank by bank: same bits in the forward copy and in the transposed
    data-gradient bank, on every bank family of the model plus shapes the vector path must refuse (row padding, odd Co, a
    master that is not 16-byte aligned, Ci not a multiple of 32)."""
    from sm3hip import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtname]
    code = ops.dtype_code(dt)
    D = torch.device(DEV)
    g = torch.Generator().manual_seed(9)
    # Co, taps, Ci, ld - K, element offset of the master inside its buffer
    shapes = [(64, 9, 64, 0, 0), (256, 1, 64, 0, 0), (512, 9, 512, 0, 0), (2048, 1, 512, 0, 0), (2048, 1, 2048, 0, 0),
              (128, 1, 2048, 0, 0), (64, 1, 147, 13, 0), (66, 9, 64, 0, 0), (33, 1, 64, 0, 0), (64, 1, 48, 0, 0), (128, 9, 128, 0, 3)]
    items, singles = [], []
    for Co, taps, Ci, pad, off in shapes:
        K = taps * Ci
        buf = torch.randn(Co * K + off, generator=g).to(D)
        w = buf[off:].view(Co, K)
        wf, wd = torch.zeros(Co, K + pad, dtype=dt, device=D), torch.zeros(Ci * taps * Co, dtype=dt, device=D)
        wf1, wd1 = torch.full_like(wf, 7.0), torch.full_like(wd, 7.0)
        items.append((w, wf, wd, Co, taps, Ci, K + pad))
        ops.weight_prep(code, w, Co, taps, Ci, wf1, K + pad, wd1)
        singles.append((wf1, wd1, w))
    table = ops.weight_prep_table(items, D)
    ops.weight_prep_batch(code, table)
    torch.cuda.synchronize()
    for (w, wf, wd, Co, taps, Ci, ld), (wf1, wd1, _) in zip(items, singles):
        assert torch.equal(wf, wf1), ("forward bank", Co, taps, Ci, ld)
        assert torch.equal(wd, wd1), ("data-gradient bank", Co, taps, Ci, ld)
        ref = w.view(Co, taps, Ci).permute(2, 1, 0).contiguous().to(dt).reshape(-1)   # wd[ci][t][co] = w[co][t][ci]
        assert torch.equal(wd, ref), ("transpose", Co, taps, Ci)


def _fwd_index(i, cases):
    """Position of case i's plain forward output in the flat result list of the full variants."""
    pos = 0
    for j, (N, Ci, Co, H, k) in enumerate(cases):
        if j == i:
            return pos
        pos += 2 + (2 if k == 1 else 0) + 4
    raise IndexError(i)
