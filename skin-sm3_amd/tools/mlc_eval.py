"""Supervised fine-tuning / evaluation of the multi-label model on MI355X -- entry point mirroring the reference's
tools/mlc_eval.py (Model :67-115 with BIASED prototypes, train :118-200, validate :203-280, freeze modes :374-388,
run.sh:49-61): loads a tools/mlc_train.py checkpoint (strict=False: the bias-free DeepCluster prototypes are replaced),
trains with the weighted cross-entropy sum / 8 against the real labels, reports AUROC "8 avg" (sm3hip.metrics).

    python tools/mlc_eval.py --data-name synthetic -a resnet50 -b 128 -lr 1e-3 --epochs 2 --mlc-proj v4 \
        --mlc-proj-dim 512 --num-heads 1 --sa-dim-ff 128 --sa-dropout 0.1 --finetune projector \
        --pretrain-path logs/mlc_train/ckp_149.pth

--finetune fc: extractor and label projectors frozen; projector: extractor frozen; all: encoder stages layer1-4 of both
backbones trainable as well (stem frozen), gradient flowing into the HIP encoders through the autograd bridge.  The heads
train on sm3hip/mlc.py in every mode.  The derm7pt dataset is out of scope: synthetic images and labels."""
import argparse
import os
import sys
import time

SCRIPT_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_PATH = os.path.split(SCRIPT_DIR)[0]
sys.path.insert(0, ROOT_PATH)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: see sm3hip/__init__.py

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from sm3hip.metrics import CLASSES_NAME, NUM_CLASSES, auc_avg  # noqa: E402
from src.models.projector import MultiLabelProjector4  # noqa: E402
from src.models.simclr import SimCLRSkinV32  # noqa: E402


class Model(nn.Module):
    """Reference mlc_eval.py:67-115: as mlc_train's Model, prototypes with bias (normal(0, 0.01) / zero init)."""

    def __init__(self, extractor, projectors, feat_dim, l2_norm, n_heads, sa_dim_ff, sa_dropout):
        super().__init__()
        self.extractor = extractor
        self.projectors = projectors
        self.mlc_sa = nn.TransformerEncoderLayer(d_model=feat_dim, nhead=n_heads, dim_feedforward=sa_dim_ff,
                                                 dropout=sa_dropout)
        self.feat_dim = feat_dim
        self.l2_norm = l2_norm
        self.prototypes = self._make_prototype()

    def _make_prototype(self):
        protos = nn.ModuleList([nn.Linear(self.feat_dim, i) for i in NUM_CLASSES])
        for layer in protos:
            layer.weight.data.normal_(mean=0.0, std=0.01)
            layer.bias.data.zero_()
        return protos

    def forward(self, derm_imgs, clinic_imgs):
        from sm3hip import mlc
        feats = torch.cat(self.extractor.extract(derm_imgs, clinic_imgs), dim=1)
        return mlc.heads_forward(self, feats)[1]


def get_parser():
    """src/utils/misc.py:get_parser + tools/mlc_eval.py:509-524 of the reference, then this build's own flags."""
    from src.utils.misc import get_parser as base_parser
    p = base_parser("SM3 multi-label eval / fine-tune (MI355X)")
    p.add_argument("--mlc-proj", type=str, default="v4")
    p.add_argument("--mlc-proj-dim", type=int, default=256)
    p.add_argument("--num-heads", type=int, default=1)
    p.add_argument("--sa-dim-ff", type=int, default=256)
    p.add_argument("--sa-dropout", type=float, default=0.1)
    p.add_argument("--arch-weights", type=str, default=None)
    p.add_argument("--extractor-proj-dim", type=int, default=128)
    p.add_argument("--num-labels", type=int, default=8)
    p.add_argument("--label-weights", type=float, nargs="*", default=[1.0] * 8)
    p.add_argument("--l2-norm", action="store_true")
    p.add_argument("--init-prototype", action="store_true")
    p.add_argument("--train-sz", type=int, default=224)
    p.add_argument("--test-sz", type=int, default=224)
    # this build (synthetic data: an epoch is a number of steps)
    p.add_argument("--steps-per-epoch", type=int, default=8)
    p.add_argument("--val-steps", type=int, default=4)
    p.set_defaults(arch="resnet50", batch_size=128, finetune="projector", pretrain_path="", log_path="./logs/mlc_eval")
    return p


def set_requires_grad(module, flag):
    for p in module.parameters():
        p.requires_grad = flag


def synthetic(bs, size, dev, gen):
    derm = torch.randn(bs, 3, size[0], size[1], device=dev, generator=gen)
    clinic = torch.randn(bs, 3, size[0], size[1], device=dev, generator=gen)
    labels = torch.stack([torch.randint(0, n, (bs,), device=dev, generator=gen) for n in NUM_CLASSES], dim=1)
    return derm, clinic, labels


def set_train_modes(evaluator, finetune):
    """tools/mlc_eval.py:124-138.  fc: extractor, projectors AND the self-attention layer in eval mode (no dropout in
    mlc_sa), only the prototypes train; projector: extractor in eval mode; all: everything in train mode."""
    evaluator.train()
    evaluator.extractor.train(finetune == "all")
    if finetune == "fc":
        evaluator.projectors.eval()
        evaluator.mlc_sa.eval()


def run_epoch(args, evaluator, criterion, optimizer, steps, gen, dev, train):
    if train:  # the reference's mode matrix, tools/mlc_eval.py:124-138
        set_train_modes(evaluator, args.finetune)
    else:
        evaluator.eval()
    preds_all, targets_all, total, t0 = [], [], 0.0, time.time()
    for _ in range(steps):
        derm, clinic, labels = synthetic(args.batch_size, args.img_sz, dev, gen)
        with torch.set_grad_enabled(train):
            outputs = evaluator(derm, clinic)
            loss = sum(args.label_weights[i] * criterion(outputs[i], labels[:, i]) for i in range(args.num_labels))
            loss = loss / args.num_labels
        if train:
            optimizer.zero_grad(set_to_none=True)
            scaler = getattr(args, "scaler", None)  # mlc_eval.py:157-170 of the reference: GradScaler(enabled=args.amp)
            if scaler is None:
                loss.backward()
                optimizer.step()
            else:
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
        total += float(loss.detach())
        preds_all.append([o.detach() for o in outputs])
        targets_all.append(labels)
    preds = [torch.cat([p[i] for p in preds_all]) for i in range(args.num_labels)]
    per, avg = auc_avg(preds, torch.cat(targets_all))
    stat = {f"AUC_{n}": float(v) for n, v in zip(CLASSES_NAME, per)}
    stat.update({"AUC_AVG": float(avg), "loss": total / steps, "pairs_per_s": steps * args.batch_size / (time.time() - t0)})
    return stat


def main(argv=None):
    args = get_parser().parse_args(argv)
    if args.data_name != "synthetic":
        raise SystemExit("only --data-name synthetic is available in this build (dataset pipeline is out of scope)")
    if args.mlc_proj != "v4" or args.num_labels != 8:
        raise SystemExit("the native head path builds --mlc-proj v4 with 8 labels (run.sh:49-61)")
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", 0)
    extractor = SimCLRSkinV32(arch=args.arch, proj_dim=args.extractor_proj_dim)
    extractor.derm_backbone.projector = None  # mlc_eval.py:339-341
    extractor.clinic_backbone.projector = None
    extractor.cross_proj = None
    from src.utils.misc import amp_dtype
    extractor.sm3_dtype = amp_dtype(args)
    args.scaler = torch.amp.GradScaler("cuda", enabled=amp_dtype(args) == torch.float16)  # mlc_eval.py:331
    feat_dim = extractor.derm_feat_dim + extractor.clinic_feat_dim
    evaluator = Model(extractor, MultiLabelProjector4(feat_dim, args.mlc_proj_dim, args.num_labels), args.mlc_proj_dim,
                      args.l2_norm, args.num_heads, args.sa_dim_ff, args.sa_dropout)
    if args.pretrain_path and os.path.isfile(args.pretrain_path):
        state = torch.load(args.pretrain_path, map_location="cpu", weights_only=False)["state_dict"]
        state = {k: v for k, v in state.items()
                 if not (k.startswith("prototypes.") and k not in evaluator.state_dict())}
        msg = evaluator.load_state_dict(state, strict=False)
        print(f"loaded pre-trained model weights from '{args.pretrain_path}' (missing keys: {sorted(msg.missing_keys)})")
    if args.init_prototype:
        evaluator.prototypes = evaluator._make_prototype()
    set_requires_grad(evaluator.extractor, False)  # mlc_eval.py:374-388
    if args.finetune == "fc":
        set_requires_grad(evaluator.projectors, False)
    elif args.finetune == "all":
        for bb in (evaluator.extractor.derm_backbone.encoder, evaluator.extractor.clinic_backbone.encoder):
            for name in ("layer1", "layer2", "layer3", "layer4"):
                set_requires_grad(getattr(bb, name), True)
    evaluator.to(dev)
    params = [p for p in evaluator.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(params, lr=args.base_lr, weight_decay=args.wd)
    criterion = nn.CrossEntropyLoss()
    gen = torch.Generator(device=dev).manual_seed(args.seed)
    os.makedirs(args.log_path, exist_ok=True)
    best, history = -1.0, []
    for epoch in range(args.epochs):
        tr = run_epoch(args, evaluator, criterion, optimizer, args.steps_per_epoch, gen, dev, True)
        va = run_epoch(args, evaluator, criterion, None, args.val_steps, gen, dev, False)
        history.append((tr, va))
        print(f"epoch {epoch}: train loss {tr['loss']:.4f} AUC_AVG {tr['AUC_AVG']:.4f} {tr['pairs_per_s']:.0f} pairs/s | "
              f"val loss {va['loss']:.4f} AUC_AVG {va['AUC_AVG']:.4f}", flush=True)
        if va["AUC_AVG"] > best:  # best by val/AUC_AVG
            best = va["AUC_AVG"]
            torch.save({"epoch": epoch + 1, "state_dict": evaluator.state_dict(), "optimizer": optimizer.state_dict()},
                       os.path.join(args.log_path, "best_finetune.pth"))
    return history


if __name__ == "__main__":
    main()
