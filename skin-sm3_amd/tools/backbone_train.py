"""SM3 self-supervised pre-training on MI355X -- entry point mirroring the reference's
tools/backbone_train.py (flags of src/utils/misc.py:106-225 + backbone_train.py:612-624 that the pre-training
recipe of run.sh:3-12 uses; one process per GPU; same checkpoint wire format, backbone_train.py:575-592).

    python tools/backbone_train.py --data-name synthetic --data-path - -a resnet50 --arch-version v32 \
        -b 512 -lr 1e-6 --temperature 0.1 --proj-dim 128 --epochs 1 --steps-per-epoch 20

Differences, stated: the derm7pt dataset and its PIL augmentation pipeline are host-side and out of scope
(SURVEY.md 2.1 #9-10), so `--data-name synthetic` generates normalised image pairs on the device; `--engine fused`
(default) runs the fused step of sm3hip.trainer.SM3Trainer, `--engine compat` runs the reference's literal loop
(model(...) -> CrossEntropyLoss -> backward -> torch.optim.AdamW, backbone_train.py:98-127) on the same kernels.
`--amp` means what the reference's means (fp16 autocast + GradScaler, backbone_train.py:27,98,125-127,480): fp16 storage
+ f16 MFMA with dynamic loss scaling, on the device in the fused engine and through torch.cuda.amp.GradScaler itself in
the compat engine; `--amp --amp-dtype bf16` selects bf16 (BASELINE.json's benchmark type, needs no loss scaling);
without `--amp` the exact-f32 MFMA mode runs.  Every flag of the reference's parser is accepted (src/utils/misc.py); the
ones without a meaning here (-j, --proj-name, --wandb, ...) are parsed, listed once at start-up, and ignored.
"""
import argparse
import os
import sys
import time
import traceback

SCRIPT_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_PATH = os.path.split(SCRIPT_DIR)[0]
sys.path.insert(0, ROOT_PATH)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: see sm3hip/__init__.py

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402
import torch.nn as nn  # noqa: E402

from src.models.simclr import SimCLRSkinV3, SimCLRSkinV32  # noqa: E402


def get_parser():
    """src/utils/misc.py:get_parser + tools/backbone_train.py:612-624 of the reference, then this build's own flags."""
    from src.utils.misc import get_parser as base_parser
    p = base_parser("SM3 SSL pre-training (MI355X)")
    p.add_argument("--arch-version", default="v3", type=str, choices=["v3", "v311", "v312", "v32", "v321", "v322"])
    p.add_argument("--arch-weights", default=None, type=str)
    p.add_argument("--ft-lr", default=1e-3, type=float, help="finetune learning rate (unused by the reference's loop too)")
    p.add_argument("--proj-dim", default=128, type=int)
    p.add_argument("--temperature", default=0.5, type=float)
    p.add_argument("--modality-weights", nargs=2, type=float, default=[1.0, 1.0])
    p.add_argument("--num-labels", type=int, default=8)
    p.add_argument("--label-weights", type=float, nargs="*", default=[1.0] * 8)
    p.add_argument("--use-checkpoint", action="store_true")
    # this build
    p.add_argument("--steps-per-epoch", default=100, type=int, help="synthetic data only")
    p.add_argument("--synthetic-kind", default="noise", choices=["noise", "latent"],
                   help="synthetic data: independent noise (throughput) or learnable latent-pattern pairs")
    p.add_argument("--global-negatives", action="store_true",
                   help="extension (not reference behaviour): NT-Xent against the all-gathered projections of all ranks; "
                        "world * 2 * batch and --proj-dim must be multiples of 32")
    p.add_argument("--gpu-augment", action="store_true",
                   help="synthetic uint8 source images + the reference's augmentation chain on the GPU instead of "
                        "ready-made normalised tensors")
    p.add_argument("--engine", default="fused", choices=["fused", "compat"])
    p.set_defaults(arch="resnet50", port=29533)
    return p


STYLE = {"v3": 0, "v32": 0, "v311": 1, "v321": 1, "v312": 2, "v322": 2}


def synthetic_batch(bs, size, device, gen, kind="noise"):
    """([derm view 0, view 1], [clinic view 0, view 1]) of normalised images generated on the device.
    "noise": independent N(0,1) images (throughput runs: nothing to learn).  "latent": every sample has a random
    low-frequency pattern that all four of its images show (the clinical ones colour-mixed and mirrored), under
    independent pixel noise and contrast jitter -- positives are identifiable, so the loss can fall."""
    if kind == "noise":
        mk = lambda: torch.randn(bs, 3, size[0], size[1], device=device, generator=gen)
        return [mk(), mk()], [mk(), mk()]
    z = torch.randn(bs, 3, 6, 6, device=device, generator=gen)
    base = torch.nn.functional.interpolate(z, size=tuple(size), mode="bilinear", align_corners=False) * 1.5
    mix = torch.tensor([[0.6, 0.3, 0.1], [0.2, 0.5, 0.3], [0.1, 0.2, 0.7]], device=device)
    other = torch.einsum("dc,bchw->bdhw", mix, base).flip(-1)

    def view(img):
        gain = 1.0 + 0.2 * torch.randn(bs, 1, 1, 1, device=device, generator=gen)
        return (img * gain + 0.5 * torch.randn(img.shape, device=device, generator=gen)).contiguous()

    return [view(base), view(base)], [view(other), view(other)]


def main(local_rank, args):
    world = args.world_size
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{args.port}", world_size=world, rank=local_rank)
    torch.manual_seed(args.seed)
    bs = args.batch_size // world  # src/utils/misc.py:400
    if args.data_name != "synthetic":
        raise SystemExit("only --data-name synthetic is available in this build (dataset pipeline is out of scope)")
    cls = SimCLRSkinV3 if args.arch_version in ("v3", "v311", "v312") else SimCLRSkinV32
    model = cls(arch=args.arch, weights=args.arch_weights, proj_dim=args.proj_dim, temperature=args.temperature,
                use_checkpoint=args.use_checkpoint)
    from src.utils.misc import amp_dtype
    model.sm3_dtype = amp_dtype(args)
    fp16 = model.sm3_dtype == torch.float16
    if world > 1:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model = model.to(dev)
    style = STYLE[args.arch_version]
    start_epoch = 0

    if args.engine == "fused":
        from sm3hip.trainer import SM3Trainer
        trainer = SM3Trainer(model, lr=args.base_lr, weight_decay=args.wd, eps=1e-5, style=style,
                             global_negatives=args.global_negatives)
    else:
        wrapped = nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) if world > 1 else model
        optimizer = torch.optim.AdamW(wrapped.parameters(), lr=args.base_lr, weight_decay=args.wd, eps=1e-5)
        criterion = nn.CrossEntropyLoss()
        scaler = torch.amp.GradScaler("cuda", enabled=fp16)  # backbone_train.py:480

    if args.resume_path:
        ckpt = torch.load(args.resume_path, map_location=dev)
        model.load_state_dict(ckpt["state_dict"], strict=False)
        start_epoch = ckpt.get("epoch", 0)
        if args.engine == "fused" and "optimizer" in ckpt:
            trainer.load_optimizer_state_dict(ckpt["optimizer"])
            trainer.load_scaler_state_dict(ckpt.get("scaler"))
        elif args.engine == "compat" and "optimizer" in ckpt:
            optimizer.load_state_dict(ckpt["optimizer"])
            if ckpt.get("scaler"):
                scaler.load_state_dict(ckpt["scaler"])

    gen = torch.Generator(device=dev).manual_seed(args.seed + local_rank)
    augment = None
    if args.gpu_augment:
        # the reference's transform chain (backbone_train.py:448-466) on the GPU: decoded uint8 source images in HBM ->
        # two augmented, normalised views per modality (sm3hip/augment.py, csrc/augment.hip)
        from sm3hip.augment import SimCLRAugment
        augment = SimCLRAugment(tuple(args.img_sz), args.mean, args.std)
        aug_gen = torch.Generator().manual_seed(args.seed + 1000 + local_rank)
    os.makedirs(args.log_path, exist_ok=True)
    for epoch in range(start_epoch, args.epochs):
        model.train()
        t0, seen, running = time.time(), 0, None
        for it in range(args.steps_per_epoch):
            if augment is not None:
                src_hw = (2 * args.img_sz[0] + 14, 3 * args.img_sz[1] + 46)  # ~ derm7pt's 462 x 718 at 224
                d_src = torch.randint(0, 256, (bs,) + src_hw + (3,), device=dev, generator=gen, dtype=torch.uint8)
                c_src = torch.randint(0, 256, (bs,) + src_hw + (3,), device=dev, generator=gen, dtype=torch.uint8)
                derm, clinic = augment(d_src, aug_gen), augment(c_src, aug_gen)
            else:
                derm, clinic = synthetic_batch(bs, args.img_sz, dev, gen, args.synthetic_kind)
            if args.engine == "fused":
                loss = trainer.step(derm, clinic)
            else:
                outputs = wrapped(derm, clinic, style)
                w = 0.25 if style == 2 else 0.5
                loss = criterion(*outputs[0]) + criterion(*outputs[1]) + sum(w * criterion(*o) for o in outputs[2])
                optimizer.zero_grad(set_to_none=True)
                scaler.scale(loss).backward()  # backbone_train.py:125-127 (identity when not fp16)
                scaler.step(optimizer)
                scaler.update()
            seen += bs * world
            if local_rank == 0 and it % args.print_freq == 0:
                running = float(loss)  # the only host sync, every print_freq steps
                dt = time.time() - t0
                print(f"Train epoch: [{epoch}][{it}/{args.steps_per_epoch}] Loss {running:.4f} "
                      f"{seen / max(dt, 1e-9):.1f} pairs/s", flush=True)
        if local_rank == 0:
            state = {"epoch": epoch + 1, "state_dict": model.state_dict(),
                     "optimizer": trainer.optimizer_state_dict() if args.engine == "fused" else optimizer.state_dict(),
                     "scaler": trainer.scaler_state_dict() if args.engine == "fused" else scaler.state_dict()}
            path = os.path.join(args.log_path, "checkpoint.pth.tar")
            torch.save(state, path)
            if (epoch + 1) % args.save_freq == 0:
                torch.save(state, os.path.join(args.log_path, f"ckp_{epoch + 1}.pth"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = get_parser()
    args = parser.parse_args()
    from src.utils.misc import describe_ignored
    if describe_ignored(args, parser):
        print("accepted for compatibility, without effect in this build:", " ".join(describe_ignored(args, parser)), flush=True)
    args.world_size = int(os.environ.get("SM3_WORLD_SIZE", torch.cuda.device_count()))
    try:
        if args.world_size > 1:
            mp.spawn(main, nprocs=args.world_size, args=(args,))
        else:
            main(0, args)
    except Exception:
        os.makedirs(args.log_path, exist_ok=True)
        with open(os.path.join(args.log_path, "error.log"), "a") as f:
            f.write(traceback.format_exc())
        raise
