"""Linear probe / fine-tune of the SSL backbones on MI355X -- entry point mirroring the reference's
tools/backbone_eval.py (train :65-142, validate :145-212, checkpoint key split :278-296, run.sh:15-28).

    python tools/backbone_eval.py --data-name synthetic --data-path - -a resnet50 -b 128 -lr 1e-3 \
        --finetune fc --pretrain-path logs/backbone/ckp_50.pth --epochs 2 --steps-per-epoch 10

`--finetune fc`: encoders frozen and in eval mode (one fused conv+BN+ReLU kernel per layer), the 8 heads trained
with AdamW on the weighted cross-entropy sum/8; AUROC "8 avg" (sm3hip.metrics.auc_avg) on the validation pass.
Any other value fine-tunes everything through the autograd bridge.  The derm7pt dataset is out of scope: synthetic
images and labels (see tools/backbone_train.py).
"""
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
from src.models.baseline import Baseline  # noqa: E402


def get_parser():
    """src/utils/misc.py:get_parser + tools/backbone_eval.py:434-440 of the reference, then this build's own flags."""
    from src.utils.misc import get_parser as base_parser
    p = base_parser("SM3 linear probe / fine-tune (MI355X)")
    p.add_argument("--arch-weights", type=str, default=None)
    p.add_argument("--num-labels", type=int, default=8)
    p.add_argument("--label-weights", type=float, nargs="*", default=[1.0] * 8)
    # this build (synthetic data: an epoch is a number of steps)
    p.add_argument("--steps-per-epoch", default=8, type=int)
    p.add_argument("--val-steps", default=4, type=int)
    p.set_defaults(arch="resnet50", epochs=50, batch_size=128)
    return p


def load_ssl_backbones(evaluator, path):
    """Split an SSL checkpoint's keys by the derm_backbone.encoder. / clinic_backbone.encoder. prefixes
    (backbone_eval.py:278-296)."""
    state = torch.load(path, map_location="cpu")["state_dict"]
    derm, clinic = {}, {}
    for k, v in state.items():
        k = k[7:] if k.startswith("module.") else k
        if k.startswith("derm_backbone.encoder."):
            derm[k[len("derm_backbone.encoder."):]] = v
        elif k.startswith("clinic_backbone.encoder."):
            clinic[k[len("clinic_backbone.encoder."):]] = v
    evaluator.derm_backbone.load_state_dict(derm)
    evaluator.clinic_backbone.load_state_dict(clinic)


def synthetic(bs, size, dev, gen):
    derm = torch.randn(bs, 3, size[0], size[1], device=dev, generator=gen)
    clinic = torch.randn(bs, 3, size[0], size[1], device=dev, generator=gen)
    labels = torch.stack([torch.randint(0, n, (bs,), device=dev, generator=gen) for n in NUM_CLASSES], dim=1)
    return derm, clinic, labels


def run_epoch(args, evaluator, criterion, optimizer, steps, gen, dev, train):
    if train and args.finetune != "fc":
        evaluator.train()
    else:
        evaluator.eval()
    all_preds, all_targets, total, t0 = [], [], 0.0, time.time()
    for it in range(steps):
        derm, clinic, labels = synthetic(args.batch_size, args.img_sz, dev, gen)
        with torch.set_grad_enabled(train):
            outputs = evaluator([derm, clinic])
            loss = sum(args.label_weights[i] * criterion(outputs[i], labels[:, i]) for i in range(args.num_labels))
            loss = loss / args.num_labels
        if train:
            optimizer.zero_grad(set_to_none=True)
            scaler = getattr(args, "scaler", None)  # backbone_eval.py:100-112 of the reference: GradScaler(enabled=args.amp)
            if scaler is None:
                loss.backward()
                optimizer.step()
            else:
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
        total += float(loss.detach())
        all_preds.append([o.detach() for o in outputs])
        all_targets.append(labels)
    preds = [torch.cat([p[i] for p in all_preds]) for i in range(args.num_labels)]
    per, avg = auc_avg(preds, torch.cat(all_targets))
    stat = {f"AUC_{n}": float(v) for n, v in zip(CLASSES_NAME, per)}
    stat.update({"AUC_AVG": float(avg), "loss": total / steps,
                 "pairs_per_s": steps * args.batch_size / (time.time() - t0)})
    return stat


def main():
    parser = get_parser()
    args = parser.parse_args()
    from src.utils.misc import amp_dtype, describe_ignored
    if describe_ignored(args, parser):
        print("accepted for compatibility, without effect in this build:", " ".join(describe_ignored(args, parser)), flush=True)
    if args.data_name != "synthetic":
        raise SystemExit("only --data-name synthetic is available in this build (dataset pipeline is out of scope)")
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", 0)
    evaluator = Baseline(args.arch, args.arch_weights)
    if args.pretrain_path and os.path.isfile(args.pretrain_path):
        load_ssl_backbones(evaluator, args.pretrain_path)
        print(f"loaded pre-trained model weights from '{args.pretrain_path}'")
    if args.finetune == "fc":
        evaluator.freeze_backbone()
    for m in (evaluator.derm_backbone, evaluator.clinic_backbone):
        m.sm3_dtype = amp_dtype(args)
    args.scaler = torch.amp.GradScaler("cuda", enabled=amp_dtype(args) == torch.float16)  # backbone_eval.py:271
    evaluator.to(dev)
    params = [p for p in evaluator.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(params, lr=args.base_lr, weight_decay=args.wd)
    criterion = nn.CrossEntropyLoss()
    gen = torch.Generator(device=dev).manual_seed(args.seed)
    best = -1.0
    os.makedirs(args.log_path, exist_ok=True)
    for epoch in range(args.epochs):
        tr = run_epoch(args, evaluator, criterion, optimizer, args.steps_per_epoch, gen, dev, True)
        va = run_epoch(args, evaluator, criterion, None, args.val_steps, gen, dev, False)
        print(f"epoch {epoch}: train loss {tr['loss']:.4f} AUC_AVG {tr['AUC_AVG']:.4f} {tr['pairs_per_s']:.0f} pairs/s | "
              f"val loss {va['loss']:.4f} AUC_AVG {va['AUC_AVG']:.4f} {va['pairs_per_s']:.0f} pairs/s", flush=True)
        if va["AUC_AVG"] > best:  # best by val/AUC_AVG (backbone_eval.py:386,405-411)
            best = va["AUC_AVG"]
            torch.save({"epoch": epoch + 1, "state_dict": evaluator.state_dict(), "optimizer": optimizer.state_dict()},
                       os.path.join(args.log_path, "best_linear.pth"))


if __name__ == "__main__":
    main()
