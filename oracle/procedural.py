"""TEST INFRASTRUCTURE ONLY -- seeded, platform-independent weights and inputs.

Weights are never stored in fixtures: both the golden generator (which loads them into the
reference model) and the tests (which load them into the oracle and into the HIP path)
regenerate them from ``(key, seed)`` with numpy's MT19937 ``RandomState``, whose streams are
identical on every platform.  Shapes follow the reference's modules:
``src/models/resnet.py:177-290`` (ResNet-50 = Bottleneck [3,4,6,3]) and
``src/models/simclr.py:17-27,31-52,250-267,399-413``.
"""
import zlib
from collections import OrderedDict

import numpy as np

RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (planes, blocks, stride)


def _bn_keys(prefix, c, affine=True):
    out = []
    if affine:
        out += [(prefix + ".weight", (c,)), (prefix + ".bias", (c,))]
    out += [
        (prefix + ".running_mean", (c,)),
        (prefix + ".running_var", (c,)),
        (prefix + ".num_batches_tracked", ()),
    ]
    return out


def resnet50_spec(prefix=""):
    """(key, shape) list in torch ``state_dict()`` order for the fc=Identity encoder."""
    spec = [(prefix + "conv1.weight", (64, 3, 7, 7))] + _bn_keys(prefix + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS, start=1):
        for b in range(blocks):
            p = f"{prefix}layer{li}.{b}."
            spec += [(p + "conv1.weight", (planes, inplanes, 1, 1))] + _bn_keys(p + "bn1", planes)
            spec += [(p + "conv2.weight", (planes, planes, 3, 3))] + _bn_keys(p + "bn2", planes)
            spec += [(p + "conv3.weight", (planes * 4, planes, 1, 1))] + _bn_keys(p + "bn3", planes * 4)
            if b == 0:
                spec += [(p + "downsample.0.weight", (planes * 4, inplanes, 1, 1))]
                spec += _bn_keys(p + "downsample.1", planes * 4)
                inplanes = planes * 4
    return spec


def projector_spec(prefix, in_dim=2048, proj_dim=128):
    spec = [(prefix + "0.weight", (in_dim, in_dim))] + _bn_keys(prefix + "1", in_dim)
    spec += [(prefix + "3.weight", (in_dim, in_dim))] + _bn_keys(prefix + "4", in_dim)
    spec += [(prefix + "6.weight", (proj_dim, in_dim))] + _bn_keys(prefix + "7", proj_dim, affine=False)
    return spec


def sm3_v32_spec(proj_dim=128):
    """The 700-entry state_dict layout of SimCLRSkinV32('resnet50')."""
    spec = []
    for bb in ("derm_backbone.", "clinic_backbone."):
        spec += resnet50_spec(bb + "encoder.")
        spec += projector_spec(bb + "projector.", 2048, proj_dim)
    spec += projector_spec("cross_proj.0.", 2048, proj_dim)
    spec += projector_spec("cross_proj.1.", 2048, proj_dim)
    return spec


def baseline_spec():
    """state_dict layout of the linear-probe model Baseline('resnet50') (src/models/baseline.py:60-96)."""
    spec = []
    for bb in ("derm_backbone.", "clinic_backbone."):
        spec += resnet50_spec(bb)
    for i, n in enumerate((5, 3, 2, 3, 3, 3, 3, 2)):
        spec += [(f"classifier.{i}.weight", (n, 4096)), (f"classifier.{i}.bias", (n,))]
    return spec


def inference_model_spec(proj_dim=512, sa_dim_ff=128):
    """state_dict layout of the multi-label model of inference.py (Extractor + MultiLabelProjector + one
    TransformerEncoderLayer + 8 prototype heads; reference inference.py:16-96)."""
    spec = []
    for bb in ("extractor.derm_backbone.", "extractor.clinic_backbone."):
        spec += resnet50_spec(bb)
    for i in range(8):
        spec += [(f"projectors.projectors.{i}.0.weight", (proj_dim, 4096)), (f"projectors.projectors.{i}.0.bias", (proj_dim,))]
    d = proj_dim
    spec += [("mlc_sa.self_attn.in_proj_weight", (3 * d, d)), ("mlc_sa.self_attn.in_proj_bias", (3 * d,)),
             ("mlc_sa.self_attn.out_proj.weight", (d, d)), ("mlc_sa.self_attn.out_proj.bias", (d,)),
             ("mlc_sa.linear1.weight", (sa_dim_ff, d)), ("mlc_sa.linear1.bias", (sa_dim_ff,)),
             ("mlc_sa.linear2.weight", (d, sa_dim_ff)), ("mlc_sa.linear2.bias", (d,)),
             ("mlc_sa.norm1.weight", (d,)), ("mlc_sa.norm1.bias", (d,)),
             ("mlc_sa.norm2.weight", (d,)), ("mlc_sa.norm2.bias", (d,))]
    for i, n in enumerate((5, 3, 2, 3, 3, 3, 3, 2)):
        spec += [(f"prototypes.{i}.weight", (n, d)), (f"prototypes.{i}.bias", (n,))]
    return spec


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_tensor(key, shape, seed=0):
    """Deterministic value for one state_dict entry (numpy, float32 / int64)."""
    r = _rng(key, seed)
    if key.endswith("num_batches_tracked"):
        return np.zeros((), dtype=np.int64)
    if key.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if key.endswith("running_var"):
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if len(shape) == 1 and key.endswith(".weight"):  # BN gamma: non-trivial on purpose
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if len(shape) == 1 and key.endswith("bias"):  # BN beta, Linear / LayerNorm / attention biases
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:  # conv: Kaiming normal, fan_out, relu  (resnet.py:227-229)
        fan_out = shape[0] * shape[2] * shape[3]
        std = np.sqrt(2.0 / fan_out)
        return (std * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 2:  # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
        bound = 1.0 / np.sqrt(shape[1])
        return r.uniform(-bound, bound, size=shape).astype(np.float32)
    raise ValueError(key)


def make_state_dict(spec=None, seed=0):
    spec = spec if spec is not None else sm3_v32_spec()
    return OrderedDict((k, fill_tensor(k, s, seed)) for k, s in spec)


def make_images(batch, size, seed, tag):
    """One view's batch [B,3,H,W] float32: image-like (per-sample colour mean + smooth gradient
    + noise) so that cross-sample feature spread is realistic (SURVEY.md 8c conditioning)."""
    r = _rng("img:" + tag, seed)
    h = w = size
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    out = np.empty((batch, 3, h, w), dtype=np.float32)
    for b in range(batch):
        mean = r.standard_normal(3) * 0.8
        gx, gy = r.standard_normal(3) * 0.6, r.standard_normal(3) * 0.6
        for c in range(3):
            out[b, c] = mean[c] + gx[c] * xx + gy[c] * yy
        out[b] += 0.5 * r.standard_normal((3, h, w))
    return out


def make_pair_batch(batch, size, seed):
    """(derm_imgs, clinic_imgs): two lists of two views, as the loader yields them
    (tools/backbone_train.py:85-92)."""
    derm = [make_images(batch, size, seed, "derm0"), make_images(batch, size, seed, "derm1")]
    clinic = [make_images(batch, size, seed, "clinic0"), make_images(batch, size, seed, "clinic1")]
    return derm, clinic
