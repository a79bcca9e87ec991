"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SM3 pre-training hot path.

A functional (no nn.Module) torch-CPU restatement of what the reference computes on the path
``tools/backbone_train.py:98-127``: dual ResNet-50 encoders, BN-MLP projectors, in-modal and
cross-modal NT-Xent logits, loss composition, AdamW.  It runs in fp32 or fp64.  The
arithmetic the reference delegates to PyTorch (conv2d / linear / matmul) is delegated to the
same torch CPU ops here; everything the reference gets from nn.Module state machines
(train-mode BatchNorm with running statistics, SyncBatchNorm's global statistics, the
mask/select logits layout, AdamW) is written out explicitly so that the HIP kernels can be
checked stage by stage.

Pinning: ``tests/test_oracle_golden.py`` checks this file against ``tests/golden/*.npz``,
which ``oracle/gen_golden.py`` produced by importing the reference's own Python.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .procedural import RESNET50_LAYERS

BN_EPS = 1e-5  # nn.BatchNorm default, src/models/resnet.py:192,211
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# BatchNorm (train / eval), with the SyncBatchNorm global-statistics hook
# --------------------------------------------------------------------------------------
def batchnorm(x, P, B, prefix, training, affine=True, stat_reduce=None):
    """nn.BatchNorm2d / BatchNorm1d (src/models/resnet.py:211,145-149,261; simclr.py:20-26).

    Train mode: per-channel mean and *biased* variance over every non-channel axis of this
    call's batch normalise the input; running_mean/var get the momentum-0.1 update with the
    *unbiased* variance; num_batches_tracked += 1.  ``stat_reduce(t)`` (SyncBatchNorm,
    tools/backbone_train.py:510) sums a [2C+1] vector (sum x, sum x^2, count) over ranks so the
    statistics are those of the global batch.
    """
    c = x.shape[1]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, c] + [1] * (x.dim() - 2)
    if training:
        n = x.numel() // c
        cnt = torch.tensor([float(n)], dtype=x.dtype)
        if stat_reduce is not None:
            # SyncBatchNorm: statistics of the global batch from summed (sum x, sum x^2, count)
            packed = stat_reduce(torch.cat([x.sum(dim=dims), (x * x).sum(dim=dims), cnt]))
            s1, s2, cnt = packed[:c], packed[c : 2 * c], packed[2 * c :]
            mean = s1 / cnt
            var = s2 / cnt - mean * mean  # biased
        else:
            # single process: the two-pass form (ATen's CPU kernel rounding)
            mean = x.mean(dim=dims)
            var = ((x - mean.view(shape)) ** 2).mean(dim=dims)
        with torch.no_grad():
            nn_ = float(cnt.item())
            unbiased = var.detach() * (nn_ / max(nn_ - 1.0, 1.0))
            B[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(B[prefix + ".running_mean"].dtype))
            B[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased.to(B[prefix + ".running_var"].dtype))
            B[prefix + ".num_batches_tracked"] += 1
    else:
        mean = B[prefix + ".running_mean"].to(x.dtype)
        var = B[prefix + ".running_var"].to(x.dtype)
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    if affine:
        y = y * P[prefix + ".weight"].view(shape) + P[prefix + ".bias"].view(shape)
    return y


# --------------------------------------------------------------------------------------
# ResNet-50 encoder (fc = Identity)
# --------------------------------------------------------------------------------------
def bottleneck(x, P, B, p, stride, training, stat_reduce=None, taps=None):
    """Bottleneck.forward, src/models/resnet.py:154-174 (stride on the 3x3: ':146')."""
    bn = lambda t, name: batchnorm(t, P, B, p + name, training, True, stat_reduce)
    out = F.conv2d(x, P[p + "conv1.weight"])
    out = F.relu(bn(out, "bn1"))
    out = F.conv2d(out, P[p + "conv2.weight"], stride=stride, padding=1)
    out = F.relu(bn(out, "bn2"))
    out = F.conv2d(out, P[p + "conv3.weight"])
    out = bn(out, "bn3")
    if (p + "downsample.0.weight") in P:
        identity = F.conv2d(x, P[p + "downsample.0.weight"], stride=stride)
        identity = bn(identity, "downsample.1")
    else:
        identity = x
    out = F.relu(out + identity)
    if taps is not None:
        taps[p[:-1]] = out
    return out


def resnet50_features(x, P, B, prefix, training, stat_reduce=None, taps=None):
    """ResNet._forward_impl with fc = nn.Identity (src/models/resnet.py:292-308;
    simclr.py:47-49): stem 7x7/2 -> BN -> ReLU -> maxpool 3x3/2 -> 4 stages -> avgpool -> flatten."""
    out = F.conv2d(x, P[prefix + "conv1.weight"], stride=2, padding=3)
    if taps is not None:
        taps[prefix + "conv1"] = out
    out = F.relu(batchnorm(out, P, B, prefix + "bn1", training, True, stat_reduce))
    out = F.max_pool2d(out, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps[prefix + "maxpool"] = out
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS, start=1):
        for b in range(blocks):
            out = bottleneck(out, P, B, f"{prefix}layer{li}.{b}.", stride if b == 0 else 1,
                             training, stat_reduce, taps)
    return out.mean(dim=(2, 3))  # AdaptiveAvgPool2d(1) + flatten


# --------------------------------------------------------------------------------------
# Projector and NT-Xent logits
# --------------------------------------------------------------------------------------
def projector(x, P, B, prefix, training, stat_reduce=None):
    """make_projector, src/models/simclr.py:17-27: Linear-BN-ReLU-Linear-BN-ReLU-Linear-BN(affine=False),
    all Linear bias-free."""
    h = F.linear(x, P[prefix + "0.weight"])
    h = F.relu(batchnorm(h, P, B, prefix + "1", training, True, stat_reduce))
    h = F.linear(h, P[prefix + "3.weight"])
    h = F.relu(batchnorm(h, P, B, prefix + "4", training, True, stat_reduce))
    h = F.linear(h, P[prefix + "6.weight"])
    return batchnorm(h, P, B, prefix + "7", training, False, stat_reduce)


def ntxent_index(two_b):
    """Column map of the reference's mask/select logits (src/models/simclr.py:64-88,296-320).

    Returns idx [2B, 2B-1] such that logits[i, c] = S[i, idx[i, c]] / T: column 0 is the
    positive p = (i + B) mod 2B, columns 1.. are row i of S with columns {i, p} removed in
    ascending j.
    """
    b = two_b // 2
    rows = []
    for i in range(two_b):
        p = (i + b) % two_b
        rows.append([p] + [j for j in range(two_b) if j != i and j != p])
    return torch.tensor(rows, dtype=torch.long)


def ntxent_logits(z, temperature):
    """F.normalize(dim=1) (eps 1e-12) -> S = Z Z^T -> reference logits layout -> / temperature.
    src/models/simclr.py:62-88 (in-modal), :294-320 (cross-modal)."""
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(1e-12)
    s = zn @ zn.t()
    logits = torch.gather(s, 1, ntxent_index(z.shape[0])) / temperature
    labels = torch.zeros(z.shape[0], dtype=torch.long)
    return logits, labels


def ntxent_loss_closed_form(z, temperature):
    """mean_i[ -S_ip/T + log sum_{j != i} exp(S_ij/T) ]  == CrossEntropyLoss(ntxent_logits(z))
    (tools/backbone_train.py:531 applied at :101-102,119-120)."""
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(1e-12)
    s = (zn @ zn.t()) / temperature
    n = z.shape[0]
    b = n // 2
    pos = s[torch.arange(n), (torch.arange(n) + b) % n]
    s = s.masked_fill(torch.eye(n, dtype=torch.bool), float("-inf"))
    return (torch.logsumexp(s, dim=1) - pos).mean()


def cross_entropy_zero_label(logits):
    """nn.CrossEntropyLoss() with all-zero labels (tools/backbone_train.py:531)."""
    return (torch.logsumexp(logits, dim=1) - logits[:, 0]).mean()


# --------------------------------------------------------------------------------------
# SimCLR branch / SM3 model
# --------------------------------------------------------------------------------------
def simclr_forward(x1, x2, P, B, prefix, temperature, training, stat_reduce=None, taps=None):
    """SimCLR.forward, src/models/simclr.py:54-91: the two views go through the encoder
    *separately* (BN statistics per view), the projector sees cat([f1, f2])."""
    z, feats = simclr_projections(x1, x2, P, B, prefix, training, stat_reduce, taps)
    return ntxent_logits(z, temperature), feats


def simclr_projections(x1, x2, P, B, prefix, training, stat_reduce=None, taps=None):
    """The part of SimCLR.forward before the similarity matrix (simclr.py:58-61): z [2B, proj_dim]."""
    f1 = resnet50_features(x1, P, B, prefix + "encoder.", training, stat_reduce, taps)
    f2 = resnet50_features(x2, P, B, prefix + "encoder.", training, stat_reduce)
    z = projector(torch.cat([f1, f2], dim=0), P, B, prefix + "projector.", training, stat_reduce)
    return z, (f1, f2)


def cal_logits(f1, f2, P, B, proj1, proj2, temperature, training, stat_reduce=None):
    """SimCLRSkinV3._cal_logits, src/models/simclr.py:290-322: each projector call sees its own
    B rows (BN1d statistics over B, not 2B)."""
    z = torch.cat([projector(f1, P, B, proj1, training, stat_reduce),
                   projector(f2, P, B, proj2, training, stat_reduce)], dim=0)
    return ntxent_logits(z, temperature)


def sm3_v32_projections(P, B, derm_imgs, clinic_imgs, style, training=True, stat_reduce=None,
                        cross=("cross_proj.0.", "cross_proj.1.")):
    """Projector outputs of every loss term of SimCLRSkinV32.forward, in order derm, clinic, cross...;
    each [2B, proj_dim] (rows: first half = first projector's B rows)."""
    zd, df = simclr_projections(derm_imgs[0], derm_imgs[1], P, B, "derm_backbone.", training, stat_reduce)
    zc, cf = simclr_projections(clinic_imgs[0], clinic_imgs[1], P, B, "clinic_backbone.", training, stat_reduce)
    pairs = {0: [(0, 0), (1, 1)], 1: [(0, 1), (1, 0)], 2: [(0, 0), (0, 1), (1, 0), (1, 1)]}[style]
    zs = [zd, zc]
    for a, b in pairs:
        zs.append(torch.cat([projector(df[a], P, B, cross[0], training, stat_reduce),
                             projector(cf[b], P, B, cross[1], training, stat_reduce)], dim=0))
    return zs


def sm3_v32_forward(P, B, derm_imgs, clinic_imgs, style, temperature, training=True,
                    stat_reduce=None, taps=None, cross=("cross_proj.0.", "cross_proj.1.")):
    """SimCLRSkinV32.forward, src/models/simclr.py:415-482.  cross=("cross_proj.", "cross_proj.") gives
    SimCLRSkinV3.forward (one shared cross projector, :324-391)."""
    derm_outs, df = simclr_forward(derm_imgs[0], derm_imgs[1], P, B, "derm_backbone.", temperature,
                                   training, stat_reduce, taps)
    clinic_outs, cf = simclr_forward(clinic_imgs[0], clinic_imgs[1], P, B, "clinic_backbone.",
                                     temperature, training, stat_reduce)
    pairs = {0: [(0, 0), (1, 1)], 1: [(0, 1), (1, 0)], 2: [(0, 0), (0, 1), (1, 0), (1, 1)]}[style]
    cross = tuple(
        cal_logits(df[a], cf[b], P, B, cross[0], cross[1], temperature, training, stat_reduce)
        for a, b in pairs
    )
    return derm_outs, clinic_outs, cross, (df, cf)


def sm3_loss(outputs, style):
    """Loss composition, tools/backbone_train.py:99-121."""
    derm_outs, clinic_outs, cross = outputs[:3]
    w = 0.25 if style == 2 else 0.5
    cross_loss = sum(w * cross_entropy_zero_label(lg) for lg, _ in cross)
    return cross_entropy_zero_label(derm_outs[0]) + cross_entropy_zero_label(clinic_outs[0]) + cross_loss


def momentum_target_loss(P, B, P_target, derm_imgs, clinic_imgs, style, temperature):
    """BASELINE.json north_star's "momentum-updated target encoders" as SM3Trainer(target_momentum=) defines them -- an
    EXTENSION with no reference semantics (the reference has no target network; its nearest code, the never-called queue
    helper src/utils/misc.py:629-659, defines no loss).  Every term of the loss composition (sm3_loss weights) becomes the
    symmetrised query / key pair
        w/2 * NTXent(cat[q_first_half, k_second_half]) + w/2 * NTXent(cat[k_first_half, q_second_half])
    with q the ONLINE projections (parameters P, gradient flows, BatchNorm buffers B advance) and k the TARGET network's
    projections of the same batch (parameters P_target, train-mode batch statistics, no gradient, buffers untouched).
    Both halves of a concatenation act as anchors and as candidates; only q receives gradient."""
    q = sm3_v32_projections(P, B, derm_imgs, clinic_imgs, style, training=True)
    with torch.no_grad():
        Bt = {k: v.clone() for k, v in B.items()}
        k = sm3_v32_projections(P_target, Bt, derm_imgs, clinic_imgs, style, training=True)
    wc = 0.25 if style == 2 else 0.5
    weights = [1.0, 1.0] + [wc] * (len(q) - 2)
    loss = 0.0
    for w, zq, zk in zip(weights, q, k):
        h = zq.shape[0] // 2
        loss = loss + 0.5 * w * (ntxent_loss_closed_form(torch.cat([zq[:h], zk[h:]], 0), temperature)
                                 + ntxent_loss_closed_form(torch.cat([zk[:h], zq[h:]], 0), temperature))
    return loss


def ntxent_global_rows(z_local, z_all, offset, temperature):
    """NT-Xent of this rank's rows against the candidate rows of EVERY rank (opt-in "global negatives" mode of the build;
    not reference behaviour): mean_i[ -S_ip/T + log sum_{j != self} exp(S_ij/T) ] with S the cosines
    of the local rows (anchors) against z_all, self = offset + i, positive = offset + (i + B) % 2B."""
    R = z_local.shape[0]
    zl = z_local / z_local.norm(dim=1, keepdim=True).clamp_min(1e-12)
    za = z_all / z_all.norm(dim=1, keepdim=True).clamp_min(1e-12)
    s = zl @ za.t() / temperature
    idx = torch.arange(R)
    pos = s[idx, offset + (idx + R // 2) % R]
    mask = torch.zeros_like(s, dtype=torch.bool)
    mask[idx, offset + idx] = True
    return (torch.logsumexp(s.masked_fill(mask, float("-inf")), dim=1) - pos).mean()


def extract(P, B, derm, clinic):
    """SimCLRSkinV3.extract, src/models/simclr.py:393-396 (whatever mode the caller set; the
    callers use eval mode)."""
    return [resnet50_features(derm, P, B, "derm_backbone.encoder.", False),
            resnet50_features(clinic, P, B, "clinic_backbone.encoder.", False)]


# --------------------------------------------------------------------------------------
# Linear probe (tools/backbone_eval.py)
# --------------------------------------------------------------------------------------
def baseline_forward(P, B, derm, clinic, training=False):
    """Baseline.forward, src/models/baseline.py:98-102: two encoders, concatenated features, 8 Linear heads."""
    fd = resnet50_features(derm, P, B, "derm_backbone.", training)
    fc = resnet50_features(clinic, P, B, "clinic_backbone.", training)
    feats = torch.cat([fd, fc], dim=1)
    return [F.linear(feats, P[f"classifier.{i}.weight"], P[f"classifier.{i}.bias"]) for i in range(8)]


def transformer_encoder_layer(x, P, prefix, nhead=1, eps=1e-5):
    """nn.TransformerEncoderLayer in eval mode as inference.py:58-60 builds it (post-norm, ReLU, dropout inactive,
    batch_first=False): x [S, B, D] -> [S, B, D].  Attention over the S = 8 label tokens of every sample."""
    S, Bn, D = x.shape
    hd = D // nhead
    qkv = F.linear(x, P[prefix + "self_attn.in_proj_weight"], P[prefix + "self_attn.in_proj_bias"])  # [S,B,3D]
    q, k, v = qkv.split(D, dim=-1)

    def heads(t):  # [S,B,D] -> [B,nhead,S,hd]
        return t.reshape(S, Bn, nhead, hd).permute(1, 2, 0, 3)

    att = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(hd), dim=-1) @ heads(v)  # [B,nhead,S,hd]
    att = att.permute(2, 0, 1, 3).reshape(S, Bn, D)
    att = F.linear(att, P[prefix + "self_attn.out_proj.weight"], P[prefix + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + att, (D,), P[prefix + "norm1.weight"], P[prefix + "norm1.bias"], eps)
    ff = F.linear(F.relu(F.linear(x, P[prefix + "linear1.weight"], P[prefix + "linear1.bias"])),
                  P[prefix + "linear2.weight"], P[prefix + "linear2.bias"])
    return F.layer_norm(x + ff, (D,), P[prefix + "norm2.weight"], P[prefix + "norm2.bias"], eps)


def inference_forward(P, B, derm, clinic, l2_norm=False, nhead=1):
    """inference.py Model.forward (:79-96) in eval mode: frozen encoders -> cat [B,4096] -> 8 Linear(4096,512)
    label tokens -> one TransformerEncoderLayer over the 8 tokens -> (optional L2 norm) -> 8 prototype heads."""
    fd = resnet50_features(derm, P, B, "extractor.derm_backbone.", False)
    fc = resnet50_features(clinic, P, B, "extractor.clinic_backbone.", False)
    feats = torch.cat([fd, fc], dim=1)
    tokens = torch.stack([F.linear(feats, P[f"projectors.projectors.{i}.0.weight"], P[f"projectors.projectors.{i}.0.bias"])
                          for i in range(8)], dim=0)
    sa = transformer_encoder_layer(tokens, P, "mlc_sa.", nhead)
    if l2_norm:
        sa = F.normalize(sa, dim=-1, p=2)
    return [F.linear(sa[i % sa.shape[0]], P[f"prototypes.{i}.weight"], P[f"prototypes.{i}.bias"]) for i in range(8)]


def linear_probe_loss(outputs, labels, label_weights=(1.0,) * 8):
    """tools/backbone_eval.py:101-105: sum_i w_i * CE(out_i, labels[:, i]) / num_labels."""
    return sum(w * F.cross_entropy(o, labels[:, i]) for i, (o, w) in enumerate(zip(outputs, label_weights))) / len(outputs)


def auroc_selected(preds, targets, num_classes=(5, 3, 2, 3, 3, 3, 3, 2), cls_weights=(2, 2, 1, 2, 2, 2, 2, 1)):
    """src/utils/misc.py:299-327 with torchmetrics' multiclass_auroc(average=None) restated through
    sklearn.metrics.roc_auc_score on softmax probabilities (one-vs-rest, the class index CLS_WEIGHTS[i]);
    returns (per-label values, their mean = AUC_AVG)."""
    from sklearn.metrics import roc_auc_score
    per = []
    for i, (n, c) in enumerate(zip(num_classes, cls_weights)):
        prob = torch.softmax(preds[i].double(), dim=1)[:, c].numpy()
        per.append(float(roc_auc_score((targets[:, i].numpy() == c).astype(int), prob)))
    return per, sum(per) / len(per)


# --------------------------------------------------------------------------------------
# AdamW
# --------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-5, weight_decay=5e-2):
    """torch.optim.AdamW (tools/backbone_train.py:525-527: eps=1e-5, wd=args.wd, betas default),
    single-tensor form, in place.  ``step`` is the 1-based step count after increment."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def split_state(state_np, dtype=torch.float32, requires_grad=True):
    """numpy state_dict -> (params P with grad, buffers B)."""
    P, B = OrderedDict(), OrderedDict()
    for k, v in state_np.items():
        t = torch.from_numpy(v.copy()) if hasattr(v, "dtype") and not isinstance(v, torch.Tensor) else v.clone()
        if k.endswith(("running_mean", "running_var")):
            B[k] = t.to(dtype)
        elif k.endswith("num_batches_tracked"):
            B[k] = t.to(torch.long)
        else:
            P[k] = t.to(dtype).requires_grad_(requires_grad)
    return P, B


def train_step(P, B, derm_imgs, clinic_imgs, style, temperature, opt_state=None, lr=None,
               weight_decay=5e-2, eps=1e-5, stat_reduce=None):
    """One full reference step: forward, 4-term loss, backward, optional AdamW
    (tools/backbone_train.py:98-127 without the GradScaler, which is an identity in fp32)."""
    for p in P.values():
        p.grad = None
    outs = sm3_v32_forward(P, B, derm_imgs, clinic_imgs, style, temperature, True, stat_reduce)
    loss = sm3_loss(outs, style)
    loss.backward()
    if opt_state is not None:
        opt_state["step"] = opt_state.get("step", 0) + 1
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is None:
                    continue
                m = opt_state.setdefault("m." + k, torch.zeros_like(p))
                v = opt_state.setdefault("v." + k, torch.zeros_like(p))
                adamw_step(p, p.grad, m, v, opt_state["step"], lr, eps=eps, weight_decay=weight_decay)
    return loss.detach(), outs
