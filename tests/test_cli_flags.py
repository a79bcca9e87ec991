"""CPU: the four tools accept the reference's own command lines.

The argument lists below are the four invocations of /root/reference/run.sh (:3-12 SSL pre-training, :17-26 linear probe,
:31-41 multi-label DeepCluster training, :45-57 its evaluation) token for token, with two substitutions: the dataset is
`synthetic` (the derm7pt loader is host-side and out of scope) and the shell variables are expanded.  Every flag
of src/utils/misc.py:106-225 + each tool's own `parser.add_argument` lines parses; `--amp` means fp16 + loss scaling, as
in the reference (tools/backbone_train.py:27,98,125-127,480); the compatibility-only flags are reported, not dropped
silently.
"""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "skin-sm3_amd", "tools")

COMMON = ["-a", "resnet50", "--data-name", "synthetic", "--data-path", "./data/7PC",
          "--mean", "0.7833", "0.6712", "0.6026", "--std", "0.2139", "0.2472", "0.2571"]
RUN_SH = {
    # run.sh:3-12
    "backbone_train": ["-a", "resnet50", "--arch-version", "v32", "--data-name", "synthetic", "--data-path", "./data/7PC",
                       "--mean", "0.7833", "0.6712", "0.6026", "--std", "0.2139", "0.2472", "0.2571",
                       "--epochs", "400", "-b", "96", "-lr", "1e-6", "-j", "4", "--img-sz", "224", "224", "--num-labels", "8",
                       "--proj-dim", "128", "--temperature", "0.1", "--log-path", "./logs/backbone",
                       "--proj-name", "sm3_r50_backbone", "--arch-weights", "IMAGENET1K_V1", "--amp"],
    # run.sh:17-26
    "backbone_eval": COMMON + ["--epochs", "50", "-b", "128", "-lr", "1e-3", "-j", "4", "--img-sz", "224", "224",
                               "--num-labels", "8", "--pretrain-path", "./logs/backbone/ckp_49.pth", "--finetune", "fc",
                               "--log-path", "./logs/backbone/test_49", "--proj-name", "sm3_r50_backbone_eval", "--amp"],
    # run.sh:31-41
    "mlc_train": COMMON + ["--epochs", "150", "-b", "256", "-lr", "1e-4", "-j", "4", "--img-sz", "224", "224",
                           "--num-labels", "8", "--temperature", "1", "--mlc-proj", "v4", "--mlc-proj-dim", "512",
                           "--num-heads", "1", "--sa-dim-ff", "128", "--sa-dropout", "0.1", "--extractor-proj-dim", "128",
                           "--extractor-weights", "./logs/backbone/ckp_399.pth", "--log-path", "./logs/mlc_train",
                           "--proj-name", "SM3_MLC_train_v4_r50"],
    # run.sh:45-57
    "mlc_eval": COMMON + ["--epochs", "100", "-b", "128", "-lr", "1e-3", "-j", "4", "--img-sz", "224", "224",
                          "--num-labels", "8", "--mlc-proj", "v4", "--mlc-proj-dim", "512", "--num-heads", "1",
                          "--sa-dim-ff", "128", "--sa-dropout", "0.1", "--extractor-proj-dim", "128",
                          "--pretrain-path", "./logs/mlc_train/ckp_49.pth", "--finetune", "projector",
                          "--log-path", "./logs/mlc_train/test_49", "--proj-name", "SM3_MLC_eval_v4_r50"],
}


def _tool(name):
    spec = importlib.util.spec_from_file_location("sm3_cli_" + name, os.path.join(TOOLS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name", sorted(RUN_SH))
def test_the_reference_command_line_parses(name):
    from src.utils.misc import amp_dtype, describe_ignored
    parser = _tool(name).get_parser()
    args = parser.parse_args(RUN_SH[name])
    assert args.arch == "resnet50" and args.img_sz == [224, 224] and args.num_labels == 8 and args.workers == 4
    assert args.mean == [0.7833, 0.6712, 0.6026] and args.std == [0.2139, 0.2472, 0.2571]
    assert args.proj_name.lower().startswith("sm3_")
    ignored = describe_ignored(args, parser)
    assert "--workers" in ignored and "--proj-name" in ignored           # reported, not dropped silently
    if name == "backbone_train":
        assert (args.arch_version, args.batch_size, args.base_lr, args.epochs) == ("v32", 96, 1e-6, 400)
        assert (args.proj_dim, args.temperature, args.arch_weights) == (128, 0.1, "IMAGENET1K_V1")
    if name in ("backbone_train", "backbone_eval"):
        # the reference's --amp: fp16 autocast + GradScaler
        assert args.amp and amp_dtype(args) == torch.float16
    else:
        assert not args.amp and amp_dtype(args) == torch.float32
    if name == "mlc_train":
        assert (args.mlc_proj, args.mlc_proj_dim, args.sa_dim_ff, args.temperature) == ("v4", 512, 128, 1.0)
        assert args.extractor_weights.endswith("ckp_399.pth")
    if name == "mlc_eval":
        assert (args.finetune, args.mlc_proj_dim, args.pretrain_path) == ("projector", 512, "./logs/mlc_train/ckp_49.pth")


def test_bf16_stays_available_behind_amp_dtype():
    from src.utils.misc import amp_dtype
    parser = _tool("backbone_train").get_parser()
    args = parser.parse_args(["--data-name", "synthetic", "--data-path", "-", "--amp", "--amp-dtype", "bf16"])
    assert amp_dtype(args) == torch.bfloat16
    args = parser.parse_args(["--data-name", "synthetic", "--data-path", "-", "--amp-dtype", "bf16"])
    assert amp_dtype(args) == torch.float32                                # no --amp: the exact-f32 mode


def test_every_flag_of_the_reference_parser_is_accepted():
    """The reference's common parser (src/utils/misc.py:106-225), flag by flag, incl. short forms and arities."""
    parser = _tool("backbone_train").get_parser()
    argv = ["--data-name", "synthetic", "--data-path", "-", "--img-sz", "448", "448", "--n-classes", "7",
            "--mean", "0.5", "0.5", "0.5", "--std", "0.2", "0.2", "0.2", "--arch", "resnet50", "--finetune", "all",
            "--epochs", "1", "--batch-size", "8", "--base-lr", "1e-3", "--final-lr", "1e-6", "--momentum", "0.9",
            "--wd", "0.05", "--warmup-epochs", "1", "--start-warmup", "0", "--port", "29999",
            "--dist-url", "tcp://127.0.0.1", "--world-size", "1", "--rank", "0", "--seed", "1", "--workers", "2",
            "--save-freq", "1", "--print-freq", "1", "--amp", "--resume-path", "x.pth", "--pretrain-path", "y.pth",
            "--log-path", "./logs", "--logger-name", "n", "--tensorboard", "--wandb", "--run-group", "g",
            "--run-name", "r", "--run-tag", "a", "b", "--run-type", "train", "--comments", "c", "--proj-name", "p",
            # tools/backbone_train.py:612-624
            "--arch-version", "v32", "--arch-weights", "IMAGENET1K_V1", "--ft-lr", "1e-3", "--proj-dim", "128",
            "--temperature", "0.1", "--modality-weights", "1.0", "1.0", "--num-labels", "8", "--use-checkpoint"]
    args = parser.parse_args(argv)
    assert args.run_tag == ["a", "b"] and args.img_sz == [448, 448] and args.use_checkpoint and args.wandb
    # and the defaults are the reference's where the flag is the reference's
    d = _tool("backbone_train").get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-"])
    assert (d.arch_version, d.temperature, d.proj_dim, d.wd, d.seed, d.workers) == ("v3", 0.5, 128, 5e-2, 3407, 8)
    assert 49152 <= parser.get_default("port") or parser.get_default("port") == 29533
