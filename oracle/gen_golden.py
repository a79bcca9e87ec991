"""TEST INFRASTRUCTURE ONLY (build container) -- golden-vector generator.

Imports the reference's own Python (``/root/reference/src/models/simclr.py``) on CPU through
the metadata stub in ``ref_stub.py``, loads the procedural state_dict (``procedural.py``) into
``SimCLRSkinV32('resnet50', None, 128, 0.1)``, runs the reference's training step exactly as
``tools/backbone_train.py:98-127`` composes it (style 0, fp32 and fp64, no AMP) and writes small
fixtures to ``tests/golden/``.  The reference never travels: only inputs' seeds and expected
outputs are stored.

    python -m oracle.gen_golden            # from /root/repo
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SM3_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, ROOT)
from oracle import procedural, ref_stub  # noqa: E402


def _import_reference():
    ref_stub.install()
    sys.path.insert(0, REF)
    from src.models.simclr import SimCLRSkinV32  # the reference's class
    return SimCLRSkinV32


def _subsample(t, n=256):
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].double().numpy()


def run_case(SimCLRSkinV32, batch, size, seed, dtype, style, lr, tag, extract_rows=None, save=True):
    torch.manual_seed(0)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    state = procedural.make_state_dict(seed=seed)
    assert list(state.keys()) == list(model.state_dict().keys()), "state_dict key order differs"
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model = model.to(dtype)
    model.train()
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    derm = [torch.from_numpy(a).to(dtype) for a in derm_np]
    clinic = [torch.from_numpy(a).to(dtype) for a in clinic_np]

    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=5e-2, eps=1e-5)

    # intermediate taps (forward hooks on the reference modules; first derm view only)
    taps = {}
    enc = model.derm_backbone.encoder
    calls = {"n": 0}

    def mk(name):
        def hook(_m, _i, o):
            if name not in taps:
                taps[name] = o.detach().clone()
        return hook

    hooks = [enc.conv1.register_forward_hook(mk("conv1")), enc.maxpool.register_forward_hook(mk("maxpool"))]
    for li in range(1, 5):
        layer = getattr(enc, f"layer{li}")
        for b in range(len(layer)):
            hooks.append(layer[b].register_forward_hook(mk(f"layer{li}.{b}")))

    outputs = model(derm, clinic, style)
    for h in hooks:
        h.remove()
    w = 0.25 if style == 2 else 0.5
    cross_loss = sum(w * criterion(*o) for o in outputs[2])
    derm_loss = criterion(*outputs[0])
    clinic_loss = criterion(*outputs[1])
    loss = derm_loss + clinic_loss + cross_loss
    optimizer.zero_grad(set_to_none=True)
    loss.backward()

    out = {
        "meta": np.array([batch, size, seed, style], dtype=np.int64),
        "lr": np.array(lr),
        "loss": np.array(loss.item()),
        "loss_terms": np.array([derm_loss.item(), clinic_loss.item()] + [criterion(*o).item() for o in outputs[2]]),
        "derm_logits": outputs[0][0].detach().double().numpy(),
        "clinic_logits": outputs[1][0].detach().double().numpy(),
    }
    for i, o in enumerate(outputs[2]):
        out[f"cross_logits_{i}"] = o[0].detach().double().numpy()
        assert int(o[1].abs().sum()) == 0 and o[1].dtype == torch.long

    for name, t in taps.items():
        out["tap_sub." + name] = _subsample(t)
        out["tap_stat." + name] = np.array([t.double().mean().item(), t.double().abs().mean().item(),
                                             t.double().pow(2).sum().sqrt().item()])
        out["tap_shape." + name] = np.array(t.shape, dtype=np.int64)

    names = [k for k, _ in model.named_parameters()]
    out["grad_norm"] = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
    out["grad_sum"] = np.array([p.grad.double().sum().item() for _, p in model.named_parameters()])
    for k in ("derm_backbone.encoder.conv1.weight", "derm_backbone.encoder.bn1.weight",
              "derm_backbone.encoder.bn1.bias", "clinic_backbone.encoder.layer4.2.bn3.weight",
              "derm_backbone.encoder.layer1.0.conv1.weight",
              "derm_backbone.projector.4.bias", "cross_proj.1.1.weight"):
        out["grad_full." + k] = dict(model.named_parameters())[k].grad.double().numpy()
    for k in ("derm_backbone.encoder.layer3.1.conv2.weight", "clinic_backbone.projector.0.weight",
              "clinic_backbone.encoder.layer2.0.downsample.0.weight",
              "cross_proj.0.6.weight", "derm_backbone.encoder.layer4.0.conv2.weight"):
        out["grad_sub." + k] = _subsample(dict(model.named_parameters())[k].grad)

    # eval-mode extract (simclr.py:393-396): pre-step weights, post-forward running statistics
    # (taken before optimizer.step(): the first Adam step is sign-like, hence ill-conditioned)
    model.eval()
    with torch.no_grad():
        feats = model.extract(derm[0], clinic[0])
    out["extract_derm"] = feats[0][:extract_rows].double().numpy()  # extract_rows: first rows only (fixture size)
    out["extract_clinic"] = feats[1][:extract_rows].double().numpy()
    model.train()

    optimizer.step()
    sd = model.state_dict()
    out["post_param_norm"] = np.array([sd[k].double().norm().item() for k in names])
    out["post_param_sum"] = np.array([sd[k].double().sum().item() for k in names])
    bn_keys = [k for k in sd if k.endswith(("running_mean", "running_var"))]
    out["post_buf_sum"] = np.array([sd[k].double().sum().item() for k in bn_keys])
    out["post_buf_norm"] = np.array([sd[k].double().norm().item() for k in bn_keys])
    out["post_nbt"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")], dtype=np.int64)
    for k in ("derm_backbone.encoder.bn1.running_mean", "derm_backbone.encoder.bn1.running_var",
              "clinic_backbone.encoder.layer4.2.bn3.running_var", "cross_proj.0.7.running_mean",
              "derm_backbone.projector.1.running_var"):
        out["post_buf_full." + k] = sd[k].double().numpy()

    if not save:
        return out
    path = os.path.join(OUT, f"sm3_v32_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss={loss.item():.8f} ({os.path.getsize(path)/1024:.0f} KiB)")
    return names


def run_b32_case(cls):
    """The well-conditioned case (round 2): B = 32 pairs at 64x64 -- BatchNorm1d over 32 / 64 rows instead of 4 / 8,
    and the smallest batch that takes the engine's both-views-in-one-batch route.  Also runs the reference in fp32
    and prints its own fp32-vs-fp64 spread: the floor any fp32-accumulating implementation can be held to."""
    run_case(cls, batch=32, size=64, seed=6, dtype=torch.float64, style=0, lr=1e-3, tag="b32_s64_f64", extract_rows=8)
    g64 = np.load(os.path.join(OUT, "sm3_v32_b32_s64_f64.npz"))
    g32 = run_case(cls, batch=32, size=64, seed=6, dtype=torch.float32, style=0, lr=1e-3, tag="", extract_rows=8, save=False)
    print("reference fp32 vs fp64 at B=32: |dloss| = %.3e, max|dlogit| = %.3e, grad-norm max rel = %.3e" % (
        abs(float(g32["loss"]) - float(g64["loss"])), np.abs(g32["derm_logits"] - g64["derm_logits"]).max(),
        np.max(np.abs(g32["grad_norm"] - g64["grad_norm"]) / np.maximum(g64["grad_norm"], 1e-12))))
    for k in g64.files:
        if k.startswith(("grad_full.", "grad_sub.")):
            print("  %-70s rel L2 %.3e" % (k, np.linalg.norm(g32[k] - g64[k]) / np.linalg.norm(g64[k])))


def run_baseline_case(batch, size, seed, dtype, tag):
    """The linear-probe model of tools/backbone_eval.py (reference src/models/baseline.py): eval-mode forward and
    one weighted-CE backward through the heads (frozen encoders, --finetune fc)."""
    from src.models.baseline import Baseline  # the reference's class (timm stubbed)
    torch.manual_seed(0)
    model = Baseline("resnet50", None)
    state = procedural.make_state_dict(procedural.baseline_spec(), seed=seed)
    assert list(state.keys()) == list(model.state_dict().keys()), "Baseline state_dict key order differs"
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model = model.to(dtype).eval()
    for p in list(model.derm_backbone.parameters()) + list(model.clinic_backbone.parameters()):
        p.requires_grad = False
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    derm, clinic = torch.from_numpy(derm_np[0]).to(dtype), torch.from_numpy(clinic_np[0]).to(dtype)
    r = np.random.RandomState(seed)
    labels = torch.from_numpy(np.stack([r.randint(0, n, size=batch) for n in (5, 3, 2, 3, 3, 3, 3, 2)], axis=1)).long()
    weights = [1.0, 0.5, 2.0, 1.0, 1.0, 1.5, 1.0, 1.0]
    outputs = model([derm, clinic])
    crit = torch.nn.CrossEntropyLoss()
    loss = sum(w * crit(o, labels[:, i]) for i, (o, w) in enumerate(zip(outputs, weights))) / 8
    loss.backward()
    out = {"meta": np.array([batch, size, seed], dtype=np.int64), "labels": labels.numpy(),
           "label_weights": np.array(weights), "loss": np.array(loss.item())}
    for i, o in enumerate(outputs):
        out[f"logits_{i}"] = o.detach().double().numpy()
        out[f"grad_w_{i}"] = _subsample(model.classifier[i].weight.grad, 512)
        out[f"grad_b_{i}"] = model.classifier[i].bias.grad.double().numpy()
    path = os.path.join(OUT, f"baseline_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss={loss.item():.8f} ({os.path.getsize(path)/1024:.0f} KiB)")
    with open(os.path.join(OUT, "baseline_state_dict_keys.txt"), "w") as f:
        f.write("\n".join(model.state_dict().keys()) + "\n")


def run_inference_case(batch, size, seed, dtype, tag):
    """The multi-label model of the reference's inference.py (eval mode, random but procedural weights)."""
    import importlib
    ref_inf = importlib.import_module("inference")  # /root/reference/inference.py (imports its top-level resnet.py)
    assert ref_inf.__file__.startswith("/root/reference"), ref_inf.__file__
    torch.manual_seed(0)
    extractor = ref_inf.Extractor("resnet50")
    model = ref_inf.Model(extractor, ref_inf.MultiLabelProjector(4096, 512, 8), 512, False, 1, 128, 0.1)
    state = procedural.make_state_dict(procedural.inference_model_spec(), seed=seed)
    ref_sd = model.state_dict()
    assert list(state.keys()) == list(ref_sd.keys()), "inference Model state_dict key order differs"
    assert all(tuple(ref_sd[k].shape) == state[k].shape for k in state), "inference Model shapes differ"
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model = model.to(dtype).eval()
    derm_np, clinic_np = procedural.make_pair_batch(batch, size, seed)
    with torch.no_grad():
        preds = model(torch.from_numpy(derm_np[0]).to(dtype), torch.from_numpy(clinic_np[0]).to(dtype))
    out = {"meta": np.array([batch, size, seed], dtype=np.int64)}
    for i, o in enumerate(preds):
        out[f"pred_{i}"] = o.double().numpy()
    path = os.path.join(OUT, f"inference_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")
    with open(os.path.join(OUT, "inference_state_dict_keys.txt"), "w") as f:
        f.write("\n".join(ref_sd.keys()) + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    cls = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "b32":  # generate only the round-2 well-conditioned fixture
        run_b32_case(cls)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "inference":  # regenerate only the inference-model fixture
        run_inference_case(batch=5, size=64, seed=5, dtype=torch.float64, tag="b5_s64_f64")
        return
    names = run_case(cls, batch=4, size=64, seed=1, dtype=torch.float32, style=0, lr=1e-3, tag="b4_s64_f32")
    run_case(cls, batch=4, size=64, seed=1, dtype=torch.float64, style=0, lr=1e-3, tag="b4_s64_f64")
    run_case(cls, batch=3, size=96, seed=2, dtype=torch.float64, style=2, lr=1e-3, tag="b3_s96_style2_f64")
    run_case(cls, batch=8, size=64, seed=3, dtype=torch.float64, style=1, lr=1e-3, tag="b8_s64_style1_f64")
    run_b32_case(cls)
    run_baseline_case(batch=6, size=64, seed=4, dtype=torch.float64, tag="b6_s64_f64")
    run_inference_case(batch=5, size=64, seed=5, dtype=torch.float64, tag="b5_s64_f64")
    with open(os.path.join(OUT, "param_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    # the 700 state_dict keys = checkpoint wire format (tools/backbone_train.py:578-587)
    keys = [k for k, _ in procedural.sm3_v32_spec()]
    with open(os.path.join(OUT, "state_dict_keys.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")
    # ... and each entry's shape / dtype as the REFERENCE's model reports them
    import json
    ref_sd = cls("resnet50", None, 128, 0.1).state_dict()
    with open(os.path.join(OUT, "state_dict_shapes.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype)] for k, v in ref_sd.items()}, f)


if __name__ == "__main__":
    main()
