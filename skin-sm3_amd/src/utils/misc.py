"""The command-line surface shared by the four tools (mirror of the reference's src/utils/misc.py:106-225 `get_parser`).

Every flag the reference's parser defines is accepted here with the reference's name, short form, arity and default, so
that the command lines of /root/reference/run.sh:3-12,17-26,31-41,45-57 parse unchanged (tests/test_cli_flags.py feeds
them to the four tools).  Flags that configure things outside this build's scope -- the dataset loader's worker count,
the wandb / TensorBoard loggers, run names and tags -- are parsed and ignored; `describe_ignored(args)` names the ones a
caller actually set, and the tools print that line once, so nothing is dropped silently.

`--amp` means what the reference's means (tools/backbone_train.py:27,98,125-127,480): fp16 arithmetic with dynamic loss
scaling.  bf16 -- BASELINE.json's benchmark type, which needs no loss scaling -- is `--amp --amp-dtype bf16`.
"""
import argparse
import os
import sys

# (flags, kwargs) in the reference's order; grouped as misc.py groups them
_DATA = [
    (("--data-name",), dict(type=str, required=True)),
    (("--data-path",), dict(type=str, required=True, help="path to dataset repository ('-' with --data-name synthetic)")),
    (("--img-sz",), dict(nargs=2, type=int, default=[224, 224])),
    (("--n-classes",), dict(type=int)),
    (("--mean",), dict(nargs=3, type=float, default=[0.485, 0.456, 0.406])),
    (("--std",), dict(nargs=3, type=float, default=[0.229, 0.224, 0.225])),
]
_MODEL = [
    (("-a", "--arch"), dict(default="resnet18", type=str, help="convnet architecture")),
    (("--finetune",), dict(default="fc", type=str)),
]
_OPTIM = [
    (("--epochs",), dict(default=100, type=int)),
    (("-b", "--batch-size"), dict(default=64, type=int, help="global mini-batch size")),
    (("-lr", "--base-lr"), dict(default=1e-3, type=float)),
    (("--final-lr",), dict(type=float, default=0)),
    (("--momentum",), dict(default=0.9, type=float)),
    (("--wd",), dict(default=5e-2, type=float)),
    (("--warmup-epochs",), dict(default=10, type=int)),
    (("--start-warmup",), dict(default=0, type=float)),
]
_OTHER = [
    (("--seed",), dict(type=int, default=3407)),
    (("-j", "--workers"), dict(default=8, type=int, help="data-loading workers (no loader here: parsed, ignored)")),
    (("--save-freq",), dict(type=int, default=50)),
    (("--print-freq",), dict(type=int, default=50)),
    (("--amp",), dict(action="store_true", help="fp16 + dynamic loss scaling, as in the reference")),
    (("--resume-path",), dict(type=str, default=None)),
    (("--pretrain-path",), dict(type=str, default=None)),
    (("--log-path",), dict(type=str, default="./logs")),
    (("--logger-name",), dict(type=str, default=None)),
    (("--tensorboard",), dict(action="store_true")),
    (("--wandb",), dict(action="store_true")),
    (("--run-group",), dict(default=None, type=str)),
    (("--run-name",), dict(default=None, type=str)),
    (("--run-tag",), dict(nargs="*", default=None, type=str)),
    (("--run-type",), dict(default="train", type=str)),
    (("--comments",), dict(default="PyTorch training", type=str)),
    (("--proj-name",), dict(type=str, default="PyTorch Training")),
]
# parsed for compatibility, without effect in this build (no dataset workers, no external loggers, no lr schedule knobs
# beyond what each tool implements)
IGNORED = ("workers", "n_classes", "logger_name", "tensorboard", "wandb", "run_group", "run_name", "run_tag", "run_type",
           "comments", "proj_name", "momentum", "dist_url", "rank")


def default_port():
    """misc.py:155-159: a port derived from the uid, in [49152, 65536)."""
    uid = os.getuid() if sys.platform != "win32" else 1
    return 2 ** 15 + 2 ** 14 + hash(uid) % 2 ** 14


def get_parser(desc="PyTorch Training"):
    p = argparse.ArgumentParser(description=desc)
    for group in (_DATA, _MODEL, _OPTIM):
        for flags, kw in group:
            p.add_argument(*flags, **kw)
    p.add_argument("--port", default=default_port(), type=int, help="port for distributed training")
    p.add_argument("--dist-url", default="tcp://127.0.0.1", type=str)
    p.add_argument("--world-size", default=1, type=int, help="set automatically")
    p.add_argument("--rank", default=0, type=int, help="set automatically")
    for flags, kw in _OTHER:
        p.add_argument(*flags, **kw)
    # this build's additions, common to the four tools
    p.add_argument("--amp-dtype", default="fp16", choices=["fp16", "bf16"],
                   help="16-bit type behind --amp: fp16 + loss scaling (the reference's recipe) or bf16 (no scaling)")
    return p


def amp_dtype(args):
    """torch dtype of the encoders' arithmetic for these arguments."""
    import torch
    if not args.amp:
        return torch.float32
    return torch.float16 if args.amp_dtype == "fp16" else torch.bfloat16


def describe_ignored(args, parser):
    """Names of the compatibility-only flags the caller set to something other than their default."""
    out = []
    for name in IGNORED:
        if hasattr(args, name) and getattr(args, name) != parser.get_default(name):
            out.append("--" + name.replace("_", "-"))
    return out
