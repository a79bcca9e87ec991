"""CPU tests of the host side: module surface (names, state_dict, channels_last parameter views), engine
control flow with stubbed kernels, gradient-bucket schedule coverage, loss weights."""
import os
from collections import Counter

import pytest
import torch

from fakelib import installed


@pytest.fixture(scope="module")
def model():
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(0)
    return SimCLRSkinV32("resnet50", None, 128, 0.1)


def test_module_surface_matches_reference(model, golden_dir):
    keys = open(os.path.join(golden_dir, "state_dict_keys.txt")).read().split()
    assert list(model.state_dict().keys()) == keys
    names = open(os.path.join(golden_dir, "param_names.txt")).read().split()
    assert [n for n, _ in model.named_parameters()] == names
    assert model.derm_backbone.encoder_out_dim == 2048 and model.derm_feat_dim == 2048 and model.clinic_feat_dim == 2048
    assert isinstance(model.derm_backbone.encoder.fc, torch.nn.Identity)
    assert hasattr(model.derm_backbone.encoder, "layer4") and hasattr(model.derm_backbone.encoder, "conv1")
    assert isinstance(model.cross_proj, torch.nn.ModuleList) and len(model.cross_proj) == 2
    # SyncBatchNorm conversion (tools/backbone_train.py:510) keeps the key set
    m2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert list(m2.state_dict().keys()) == keys


def test_v3_has_shared_cross_projector():
    from src.models.simclr import SimCLRSkinV3
    m = SimCLRSkinV3("resnet50", None, 128, 0.5)
    assert isinstance(m.cross_proj, torch.nn.Sequential)
    assert sum(p.numel() for p in m.parameters()) == 81651840 - 8658944


def test_param_store_views_are_channels_last_and_survive_load_state_dict(model):
    from sm3hip.engine import ParamStore
    before = {k: v.clone() for k, v in model.state_dict().items()}
    st = ParamStore(model, torch.device("cpu"))
    after = model.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), k
    w = dict(model.named_parameters())["derm_backbone.encoder.layer1.0.conv2.weight"]
    assert tuple(w.shape) == (64, 64, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    assert st.bound()
    # flat [Cout][kh][kw][Cin] order
    flat = st.flat2d(st.flat_p, "derm_backbone.encoder.layer1.0.conv2.weight")
    assert torch.equal(flat.view(64, 3, 3, 64), w.permute(0, 2, 3, 1))
    # in-place load keeps the binding; .to()/_apply breaks it and is detected
    model.load_state_dict(before)
    assert st.bound()
    total = sum((p.numel() + 15) // 16 * 16 for p in model.parameters())
    assert st.total == total and st.flat_p.numel() == total


@pytest.mark.parametrize("linbn", [True, False])
def test_engine_dry_run_sequences_and_bucket_schedule(model, linbn):
    """One full trainer step on CPU tensors with the C ABI stubbed out: every ops wrapper's host-side size /
    dtype validation runs, and the gradient-ready notifications must cover every parameter exactly once.
    linbn: conv3 -> bn3 backward by linearity (the 16-bit default, csrc/linbn.hip) or the two-pass form."""
    from sm3hip.trainer import SM3Trainer
    with installed() as fake:
        model.sm3_dtype = torch.bfloat16
        tr = SM3Trainer(model, lr=1e-3, data_parallel=False)
        eng = tr._engine()
        eng.linbn = linbn
        x = [torch.randn(2, 3, 32, 32) for _ in range(4)]
        eng.prepare(torch.device("cpu"))
        ranges = []
        orig_backward = eng.backward

        def spy_backward(saved, dz, dfeat=None):
            installed = eng.grad_ready  # single rank: the trainer's bucket-wise AdamW

            def both(f, l):
                ranges.append((f, l))
                if installed is not None:
                    installed(f, l)
            eng.grad_ready = both
            orig_backward(saved, dz, dfeat)
            eng.grad_ready = None

        eng.backward = spy_backward
        try:
            # first step of a configuration: the schedule is VERIFIED (ranges recorded, one AdamW launch at the end, a bad
            # schedule raises before any update); from the second step on AdamW runs bucket by bucket
            tr.step(x[:2], x[2:])
            first = Counter(fake.calls)
            assert first["sm3_adamw"] == 1 and len(ranges) > 4 and tr._sched_ok is not None
            del fake.calls[:]
            del ranges[:]
            tr.step(x[:2], x[2:])
        finally:
            del eng.backward  # the engine is cached on the (module-scoped) model
        calls = Counter(fake.calls)
    # 53 convs x 4 encoder passes + 3 linears x (2 in-modal + 4 cross) projector passes
    # 212 BN2d + 18 BN1d per step.  Single rank: statistics reduction + finalize are ONE launch
    # (sm3_bn_stats_finalize) wherever the statistics come from a convolution's partial rows; the units whose statistics come
    # from the moments of their input (linbn_fwd_stats) keep the plain finalize
    n_from_moments = (48 + 2 * 16) if linbn else 0
    assert calls["sm3_bn_finalize"] + calls["sm3_bn_stats_finalize"] == 53 * 4 + 3 * 6 == 230
    assert calls["sm3_bn_finalize"] == (n_from_moments if eng.fused_stats else 230)
    nlin = 64 if linbn else 0  # 16 Bottlenecks x 4 encoder passes: conv3 -> bn3 units whose backward goes by linearity
    nds = 16 if linbn else 0   # ... and the 4 downsample conv -> BatchNorm units of every pass
    assert calls["sm3_conv_wgrad_det"] == 230 - 4 - nlin - nds and calls["sm3_stem_wgrad_bn16"] == 4 and calls["sm3_stem_image_prep"] == 4  # bf16: direct stem (csrc/stem.hip); fixed-order weight gradients (round 6)
    # per unit backward: sums + coefficients, then banks / -H / weight-gradient finish as ONE launch (round 5)
    for name in ("sm3_linbn_stats", "sm3_linbn_banks_post"):
        assert calls[name] == nlin + nds, name
    assert calls["sm3_linbn_banks"] == 0 and calls["sm3_linbn_post"] == 0
    assert calls["sm3_conv_dgrad_seg_bnfuse"] == nlin and calls["sm3_bn_act_colsum"] == nlin
    assert calls["sm3_conv_gather_gemm_seg"] == nds and calls["sm3_subsample_colsum"] == nds
    # Gram matrix of the unit's input and dz^T input per view -- plain-store split-K slabs, summed in a fixed order
    assert calls["sm3_conv_wgrad_slabs"] == 2 * (nlin + nds) and calls["sm3_linbn_moments"] == 2 * (nlin + nds)
    assert calls["sm3_conv_wgrad_cat"] == 0
    # the 12 blocks without a downsample branch also run conv3 -> bn3 -> +identity -> ReLU as ONE launch, bn3's statistics
    # from the moments of conv3's input (no join pass, no pre-BatchNorm tensor)
    nfused = 48 if linbn else 0
    assert calls["sm3_linbn_fwd_stats"] == nfused + 2 * nds and calls["sm3_conv_bn_act_fused"] == nfused
    # ... and the 4 with one run the whole join (conv3, bn3, downsample conv, its BatchNorm, add, ReLU) as one two-segment GEMM
    assert calls["sm3_conv_seg_act"] == nds and calls["sm3_linbn_scale_banks"] == nds
    if linbn:
        # no backward-apply pass for bn3 nor for the downsample BatchNorms (the stem's is fused into its weight gradient)
        assert calls["sm3_bn_bwd_apply2"] == 0 and calls["sm3_bn_bwd_apply"] == 230 - 4 - 64 - 16
    else:
        # every BatchNorm gets its backward apply: the bn3 / downsample pair of a downsample block in one dual launch
        assert calls["sm3_bn_bwd_apply2"] == 4 * 4 and calls["sm3_bn_bwd_apply"] == 230 - 2 * 16 - 4
    # ... and its forward apply, except where the consumer applies it: the 16 downsample BatchNorms inside their
    # block's join (sm3_bn_add_bn_act), the 4 stem BatchNorms inside the fused BN + ReLU + max-pool pass
    assert calls["sm3_bn_add_bn_act"] == 16 - nds and calls["sm3_bn_act"] + calls["sm3_bn_act_colsum"] == 230 - 16 - 16 - 4 - nfused
    # AdamW bucket by bucket as the gradients become final (single rank): one launch per gradient-ready notification
    assert calls["sm3_ntxent_fused"] == 4 and calls["sm3_adamw"] == len(ranges) > 4
    assert calls["sm3_stem_conv_fwd16"] == 4 and calls["sm3_stem_im2col"] == 0
    assert calls["sm3_bn_relu_maxpool_fwd"] == 4 and calls["sm3_maxpool_bn_bwd"] == 4
    assert calls["sm3_maxpool3x3s2_fwd"] == 0 and calls["sm3_maxpool3x3s2_bwd"] == 0
    # bucket coverage
    names = eng.store.names
    covered = Counter()
    for first, last in ranges:
        lo = next(i for i, n in enumerate(names) if n.startswith(first))
        hi = max(i for i, n in enumerate(names) if n.startswith(last))
        for i in range(lo, hi + 1):
            covered[names[i]] += 1
    assert set(covered) == set(names)
    assert all(v == 1 for v in covered.values()), [k for k, v in covered.items() if v != 1][:5]


def test_a_bucket_schedule_with_a_hole_raises_before_any_update(model):
    """a gradient-ready schedule that does not cover every parameter must fail on the verification step, i.e.
    BEFORE a single AdamW launch has modified the weights (it used to be detected after the bucket-wise updates)."""
    from sm3hip.trainer import SM3Trainer
    with installed() as fake:
        model.sm3_dtype = torch.bfloat16
        tr = SM3Trainer(model, lr=1e-3, data_parallel=False)
        eng = tr._engine()
        x = [torch.randn(2, 3, 32, 32) for _ in range(4)]
        orig = eng._notify
        dropped = []

        def lossy(first, last):
            if not dropped and "layer3" in first:
                dropped.append((first, last))  # one stage's notification goes missing
                return
            orig(first, last)

        eng._notify = lossy
        try:
            with pytest.raises(RuntimeError, match="do not tile"):
                tr.step(x[:2], x[2:])
        finally:
            del eng._notify
        assert dropped and Counter(fake.calls)["sm3_adamw"] == 0 and tr.step_count == 0


def test_ops_reject_bad_shapes_on_host():
    from sm3hip import ops
    with installed():
        d = ops.fwd_desc(1, 2, 8, 8, 64, 64, 3, 1, 1)
        x = torch.zeros(2, 8, 8, 64, dtype=torch.bfloat16)
        w = torch.zeros(64, 9 * 64, dtype=torch.bfloat16)
        y = torch.zeros(2, 8, 8, 64, dtype=torch.bfloat16)
        ops.conv_gemm(d, x, w, y)
        with pytest.raises(ValueError):
            ops.conv_gemm(d, x[:1].contiguous(), w, y)
        with pytest.raises(ValueError):
            ops.conv_gemm(d, x.float(), w, y)
        with pytest.raises(ValueError):
            ops.conv_gemm(d, x, w[:, :64].contiguous(), y)
        with pytest.raises(ValueError):
            ops.bn_act(1, x, torch.zeros(64), torch.zeros(64), None, True, y[:1].contiguous(), 128, 64)
        bad = ops.fwd_desc(1, 2, 8, 8, 48, 64, 1, 1, 0)  # Ci not a multiple of the 128-byte K chunk
        with pytest.raises(ValueError):
            ops.conv_gemm(bad, torch.zeros(2, 8, 8, 48, dtype=torch.bfloat16), torch.zeros(64, 48, dtype=torch.bfloat16), y)


def test_cpu_tensors_are_refused_by_the_real_wrappers():
    from sm3hip import ops
    with pytest.raises(ValueError, match="GPU"):
        ops.cast_from_f32(1, torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16))


def test_loss_weights_follow_reference():
    from sm3hip.trainer import SM3Trainer

    class M:
        _KIND = "v32"
    for style, w in ((0, 0.5), (1, 0.5), (2, 0.25)):
        tr = SM3Trainer.__new__(SM3Trainer)
        tr.style = style
        names = ["derm", "clinic"] + [f"cross{i}" for i in range(4 if style == 2 else 2)]
        ws = tr._weights(names)
        assert ws["derm"] == 1.0 and ws["clinic"] == 1.0 and all(ws[n] == w for n in names[2:])


def test_checkpoint_wire_format_round_trip(model, golden_dir, tmp_path):
    """tools/backbone_train.py:575-592 / src/utils/misc.py:462-494: what this build saves has the reference's keys,
    shapes and dtypes (conv weights OIHW), survives torch.save/torch.load, and loads back (strict) after the
    parameters were re-bound to the flat channels_last store."""
    import json
    from sm3hip.engine import ParamStore
    shapes = json.load(open(os.path.join(golden_dir, "state_dict_shapes.json")))
    ParamStore(model, torch.device("cpu"))  # parameters become views into the flat buffer
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, (shape, dtype) in shapes.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k
    path = tmp_path / "checkpoint.pth.tar"
    torch.save({"epoch": 3, "state_dict": sd, "optimizer": {}, "scaler": {}}, path)
    ck = torch.load(path, map_location="cpu")
    from src.models.simclr import SimCLRSkinV32
    fresh = SimCLRSkinV32("resnet50", None, 128, 0.1)
    fresh.load_state_dict(ck["state_dict"], strict=True)
    for (k, a), (_, b) in zip(fresh.state_dict().items(), sd.items()):
        assert torch.equal(a, b), k
    assert ck["epoch"] == 3


def test_mlc_eval_mode_matrix_matches_the_reference():
    """tools/mlc_eval.py:124-138 of the reference: --finetune fc puts extractor, projectors AND the self-attention layer in
    eval mode (no dropout anywhere, only the prototypes train); projector: only the extractor; all: nothing."""
    import importlib.util
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "skin-sm3_amd", "tools")
    spec = importlib.util.spec_from_file_location("sm3_mlc_eval_modes", os.path.join(tools, "mlc_eval.py"))
    me = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(me)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.extractor = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
            self.projectors = torch.nn.Sequential(torch.nn.Linear(4, 4))
            self.mlc_sa = torch.nn.TransformerEncoderLayer(d_model=4, nhead=1, dim_feedforward=8, dropout=0.1)
            self.prototypes = torch.nn.ModuleList([torch.nn.Linear(4, 3)])
    want = {"fc": (False, False, False, True), "projector": (False, True, True, True), "all": (True, True, True, True)}
    for mode, flags in want.items():
        m = Tiny().eval()
        me.set_train_modes(m, mode)
        got = (m.extractor.training, m.projectors.training, m.mlc_sa.training, m.prototypes.training)
        assert got == flags, (mode, got)
        assert m.mlc_sa.dropout.training == flags[2]
