"""Multi-label inference model row (inference.py): the oracle pinned to a golden generated from the
reference's own inference.py (CPU); our inference.Model -- HIP encoders with the fused conv+BN inference kernel,
heads as in the reference -- against the same golden (GPU); checkpoint key layout."""
import os

import numpy as np
import pytest
import torch


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "inference_b5_s64_f64.npz"))


def _inputs(g):
    from oracle import procedural
    batch, size, seed = [int(v) for v in g["meta"]]
    state = procedural.make_state_dict(procedural.inference_model_spec(), seed=seed)
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    return state, torch.from_numpy(derm_np[0]), torch.from_numpy(clinic_np[0])


def test_oracle_inference_model_matches_reference_golden(golden_dir):
    from oracle import sm3_oracle as O
    g = _golden(golden_dir)
    state, derm, clinic = _inputs(g)
    P, B = O.split_state(state, torch.float64, requires_grad=False)
    preds = O.inference_forward(P, B, derm.double(), clinic.double())
    for i, o in enumerate(preds):
        np.testing.assert_allclose(o.numpy(), g[f"pred_{i}"], atol=1e-9)


def test_inference_model_state_dict_is_the_reference_layout(golden_dir):
    """best_linear.pth / best_finetune.pth load unchanged (inference.py:123-127): same keys, same order."""
    import inference
    m = inference.build_model()
    with open(os.path.join(golden_dir, "inference_state_dict_keys.txt")) as f:
        ref_keys = f.read().split()
    assert list(m.state_dict().keys()) == ref_keys


@pytest.mark.gpu
def test_inference_model_matches_golden(golden_dir):
    import inference
    g = _golden(golden_dir)
    state, derm, clinic = _inputs(g)
    m = inference.build_model()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    for bb in (m.extractor.derm_backbone, m.extractor.clinic_backbone):
        bb.sm3_dtype = torch.float32
    m.cuda().eval()
    with torch.no_grad():
        preds = m(derm.cuda(), clinic.cuda())
    torch.cuda.synchronize()
    scale = max(float(np.abs(g[f"pred_{i}"]).max()) for i in range(8))
    for i, o in enumerate(preds):
        # f32 MFMA encoders + f32 heads vs the fp64 reference: LayerNorm keeps the tokens O(1), so 2e-4 of the
        # largest logit is the same bar as the linear-probe row
        np.testing.assert_allclose(o.cpu().double().numpy(), g[f"pred_{i}"], atol=2e-4 * scale + 1e-6)
    # bf16 encoders: same predictions to bf16 accuracy
    for bb in (m.extractor.derm_backbone, m.extractor.clinic_backbone):
        bb.sm3_dtype = torch.bfloat16
    with torch.no_grad():
        p16 = m(derm.cuda(), clinic.cuda())
    rel = max(float((a.double().cpu() - torch.from_numpy(g[f"pred_{i}"])).norm() / np.linalg.norm(g[f"pred_{i}"]))
              for i, a in enumerate(p16))
    assert rel < 5e-2, rel


@pytest.mark.gpu
@pytest.mark.parametrize("nhead,l2", [(1, False), (4, True)])
def test_label_heads_kernels_match_torch_modules(nhead, l2):
    """sm3hip.heads.LabelHeads (GEMMs + csrc/heads.hip) against the same nn.Modules run by PyTorch in fp64:
    label projectors, multi-head attention over the 8 tokens, both residual LayerNorms, feed-forward, L2 norm,
    prototype heads.  Odd batch on purpose."""
    import inference
    from sm3hip.heads import LabelHeads
    torch.manual_seed(3)
    m = inference.Model(torch.nn.Module(), inference.MultiLabelProjector(4096, 512, 8), 512, l2, nhead, 128, 0.1)
    for p in m.parameters():  # biases / norms away from their trivial initial values
        if p.dim() == 1:
            p.data.add_(0.3 * torch.randn_like(p))
    m.eval()
    feats = torch.randn(7, 4096)
    with torch.no_grad():
        md = m.double()
        tok = torch.stack(md.projectors(feats.double()), 0)
        sa = md.mlc_sa(tok)
        if l2:
            sa = torch.nn.functional.normalize(sa, dim=-1, p=2)
        want = [md.prototypes[i](sa[i % 8]) for i in range(8)]
    m = m.float().cuda()
    heads = LabelHeads(m)
    got = heads(feats.cuda(), torch.float32)
    scale = max(float(w.abs().max()) for w in want)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert float((a.double().cpu() - b).abs().max()) < 2e-5 * max(scale, 1.0)
    got16 = heads(feats.cuda(), torch.bfloat16)
    rel = max(float((a.double().cpu() - b).norm() / b.norm()) for a, b in zip(got16, want))
    assert rel < 4e-2, rel


@pytest.mark.gpu
def test_head_kernels_edge_shapes_and_argument_checks():
    """Fewer tokens than 8, a narrow model dimension, 8 heads; and the host-side argument checks of the wrappers."""
    from sm3hip import ops
    torch.manual_seed(0)
    B, S, D, nh = 3, 5, 128, 8
    qkv = torch.randn(B * S, 3 * D)
    q, k, v = qkv.view(B, S, 3, nh, D // nh).double().unbind(2)            # [B,S,nh,hd]
    att = torch.softmax(torch.einsum("bihd,bjhd->bhij", q, k) / (D // nh) ** 0.5, -1)
    want = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B * S, D)
    out = torch.empty(B * S, D, device="cuda")
    ops.token_attention(ops.dtype_code(torch.float32), qkv.cuda(), out, B, S, D, nh)
    assert float((out.double().cpu() - want).abs().max()) < 1e-5
    a, b = torch.randn(B * S, D), torch.randn(B * S, D)
    g, be = torch.rand(D) + 0.5, torch.randn(D)
    ln = torch.empty(B * S, D, device="cuda")
    ops.add_layernorm(ops.dtype_code(torch.float32), a.cuda(), b.cuda(), g.cuda(), be.cuda(), 1e-5, ln, B * S, D)
    ref = torch.nn.functional.layer_norm((a + b).double(), (D,), g.double(), be.double(), 1e-5)
    assert float((ln.double().cpu() - ref).abs().max()) < 1e-5
    ops.add_layernorm(ops.dtype_code(torch.float32), a.cuda(), None, g.cuda(), be.cuda(), 1e-5, ln, B * S, D)  # no residual
    ref = torch.nn.functional.layer_norm(a.double(), (D,), g.double(), be.double(), 1e-5)
    assert float((ln.double().cpu() - ref).abs().max()) < 1e-5
    with pytest.raises(ValueError):
        ops.token_attention(ops.dtype_code(torch.float32), qkv.cuda(), out, B, 9, D, nh)      # more than 8 tokens
    with pytest.raises(ValueError):
        ops.token_attention(ops.dtype_code(torch.float32), qkv, out, B, S, D, nh)              # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.add_layernorm(ops.dtype_code(torch.float32), a.cuda(), b.cuda()[:-1], g.cuda(), be.cuda(), 1e-5, ln, B * S, D)
