"""Multi-label DeepCluster-style head training on MI355X -- entry point mirroring the reference's tools/mlc_train.py
(BASELINE.json configs[3]): a (by default frozen) SM3 extractor, one Linear projector per label, one
TransformerEncoderLayer over the 8 label tokens, bias-free prototype heads; per epoch the memory bank of every label is
clustered by spherical k-means and the assignments serve as pseudo-labels for a cross-entropy (mlc_train.py:116-283).

    python tools/mlc_train.py --data-name synthetic -a resnet50 -b 256 -lr 1e-4 --epochs 3 --num-labels 8 \
        --temperature 1 --mlc-proj v4 --mlc-proj-dim 512 --num-heads 1 --sa-dim-ff 128 --sa-dropout 0.1

Everything arithmetic runs on the HIP kernels: the two ResNet-50 encoders through the sm3hip engine (eval mode with the
fused conv + BN + ReLU kernels when frozen; train mode with autograd through sm3hip.bridge under --finetune-backbone),
the heads and their training through sm3hip/mlc.py, the k-means through csrc/heads_train.hip.  The loop is the
reference's own (nn.CrossEntropyLoss on the predictions / temperature, torch.optim.AdamW on the trainable parameters).
Differences, stated: the derm7pt dataset and its PIL pipeline are out of scope, `--data-name synthetic` generates a fixed
set of learnable image pairs on the device; data parallelism gathers the memory bank with torch.distributed as the
reference does (:136-143,185-186)."""
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

from src.models.projector import MultiLabelProjector4  # noqa: E402
from src.models.simclr import SimCLRSkinV32  # noqa: E402

NUM_CLASSES = [5, 3, 2, 3, 3, 3, 3, 2]


class Model(nn.Module):
    """Reference mlc_train.py:58-90: same attributes and state_dict keys; forward on the HIP kernels."""

    def __init__(self, extractor, projectors, feat_dim, l2_norm, n_heads, sa_dim_ff, sa_dropout):
        super().__init__()
        self.extractor = extractor
        self.projectors = projectors
        self.mlc_sa = nn.TransformerEncoderLayer(d_model=feat_dim, nhead=n_heads, dim_feedforward=sa_dim_ff,
                                                 dropout=sa_dropout)
        self.prototypes = nn.ModuleList([nn.Linear(feat_dim, i, bias=False) for i in NUM_CLASSES])  # DeepCluster: no bias
        self.l2_norm = l2_norm

    def forward(self, derm_imgs, clinic_imgs):
        from sm3hip import mlc
        feats = torch.cat(self.extractor.extract(derm_imgs, clinic_imgs), dim=1)   # [B, 4096], HIP encoders
        return mlc.heads_forward(self, feats)                                       # (sa_feats [S, B, D], preds)


def get_parser():
    """src/utils/misc.py:get_parser + tools/mlc_train.py:446-457 of the reference, then this build's own flags."""
    from src.utils.misc import get_parser as base_parser
    p = base_parser("SM3 DeepCluster Training v3 (MI355X)")
    p.add_argument("--num-labels", type=int, default=8)
    p.add_argument("--extractor-proj-dim", type=int, default=128)
    p.add_argument("--extractor-weights", type=str, default=None)
    p.add_argument("--mlc-proj", type=str, default="v4")
    p.add_argument("--mlc-proj-dim", type=int, default=256)
    p.add_argument("--num-heads", type=int, default=1)
    p.add_argument("--sa-dim-ff", type=int, default=256)
    p.add_argument("--sa-dropout", type=float, default=0.1)
    p.add_argument("--temperature", type=float, default=0.1)
    p.add_argument("--l2-norm", action="store_true")
    p.add_argument("--finetune-backbone", action="store_true")
    # this build
    p.add_argument("--num-samples", type=int, default=413, help="size of the synthetic training split (derm7pt: 413)")
    p.set_defaults(arch="resnet50", batch_size=256, base_lr=1e-4, epochs=150, print_freq=10, log_path="./logs/mlc_train",
                   port=29512)
    return p


def synthetic_split(n, size, device, seed):
    """A fixed training split of `n` (derm, clinic) pairs with cluster structure: every sample belongs to one of a few
    latent prototypes, so the memory bank has something to cluster."""
    g = torch.Generator(device=device).manual_seed(seed)
    protos = torch.randn(6, 3, 6, 6, device=device, generator=g)
    which = torch.randint(0, 6, (n,), device=device, generator=g)
    z = protos[which] + 0.35 * torch.randn(n, 3, 6, 6, device=device, generator=g)
    base = torch.nn.functional.interpolate(z, size=tuple(size), mode="bilinear", align_corners=False) * 1.5
    mix = torch.tensor([[0.6, 0.3, 0.1], [0.2, 0.5, 0.3], [0.1, 0.2, 0.7]], device=device)
    other = torch.einsum("dc,bchw->bdhw", mix, base).flip(-1)
    noise = lambda t: (t + 0.3 * torch.randn(t.shape, device=device, generator=g)).contiguous()
    return noise(base), noise(other)


@torch.no_grad()
def init_memory(loader, model):
    """mlc_train.py:92-113: one pass in the model's current mode, embeddings of every label into the bank."""
    idxs, embs = [], []
    for index, derm, clinic in loader():
        outputs, _ = model(derm, clinic)
        idxs.append(index)
        embs.append(outputs)
    return torch.cat(idxs, 0).long(), torch.cat(embs, 1).contiguous()  # [N_local], [S, N_local, D]


@torch.no_grad()
def cluster_memory(args, prototype, K, local_index, local_emb, nmb_kmeans_iters=10, generator=None):
    """mlc_train.py:116-189: gather the bank on rank 0, spherical k-means there, broadcast centroids + assignments,
    centroids become the prototype weights."""
    from sm3hip import mlc
    world, rank = args.world_size, args.rank
    dev = local_emb.device
    assignments = -100 * torch.ones(len(local_index) * world, dtype=torch.long, device=dev)
    centroids = torch.empty(K, local_emb.size(1), device=dev)
    if world > 1:
        # gloo has no gather for device tensors (a rehearsal of this tool with two ranks on ONE GPU runs over gloo,
        # tests/test_round4_gpu.py): the bank then goes through host copies; RCCL gathers the device tensors directly
        via_host = dist.get_backend() == "gloo" and local_emb.is_cuda
        src_emb = local_emb.contiguous().cpu() if via_host else local_emb.contiguous()
        src_idx = local_index.cpu() if via_host else local_index
        all_emb = [torch.empty_like(src_emb) for _ in range(world)] if rank == 0 else None
        all_idx = [torch.empty_like(src_idx) for _ in range(world)] if rank == 0 else None
        dist.gather(src_emb, all_emb)
        dist.gather(src_idx, all_idx)
        if rank == 0:
            all_emb, all_idx = torch.cat(all_emb, 0).to(dev), torch.cat(all_idx, 0).to(dev)
    else:
        all_emb, all_idx = local_emb, local_index
    if rank == 0:
        centroids, a = mlc.spherical_kmeans(all_emb.contiguous(), K, nmb_kmeans_iters, generator)
        assignments[all_idx] = a
    if world > 1:
        dist.broadcast(centroids, 0)
        dist.broadcast(assignments, 0)
    prototype.weight.copy_(centroids)
    return assignments


def main(local_rank, args):
    world = args.world_size
    args.rank = local_rank
    # test knobs, as in bench.py (a 2-rank rehearsal on a one-GPU box: both ranks on device 0 over gloo)
    dev_index = int(os.environ.get("SM3_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        dist.init_process_group(os.environ.get("SM3_DIST_BACKEND", "nccl"), init_method=f"tcp://127.0.0.1:{args.port}",
                                world_size=world, rank=local_rank)
    torch.manual_seed(args.seed)
    if args.data_name != "synthetic":
        raise SystemExit("only --data-name synthetic is available in this build (dataset pipeline is out of scope)")
    if args.mlc_proj != "v4" or args.num_labels != 8:
        raise SystemExit("the native head path builds --mlc-proj v4 with 8 labels (run.sh:39-47)")
    bs = args.batch_size // world

    extractor = SimCLRSkinV32(arch=args.arch, proj_dim=args.extractor_proj_dim)
    if args.extractor_weights:
        extractor.load_state_dict(torch.load(args.extractor_weights, map_location="cpu")["state_dict"])
    extractor.derm_backbone.projector = None   # mlc_train.py:344-346
    extractor.clinic_backbone.projector = None
    extractor.cross_proj = None
    from src.utils.misc import amp_dtype
    extractor.sm3_dtype = amp_dtype(args)  # frozen encoders: no gradient passes through the 16-bit arithmetic
    if not args.finetune_backbone:
        for p in extractor.parameters():
            p.requires_grad = False
    feat_dim = extractor.derm_feat_dim + extractor.clinic_feat_dim
    model = Model(extractor, MultiLabelProjector4(feat_dim, args.mlc_proj_dim, args.num_labels), args.mlc_proj_dim,
                  args.l2_norm, args.num_heads, args.sa_dim_ff, args.sa_dropout).to(dev)
    wrapped = nn.parallel.DistributedDataParallel(model, device_ids=[dev_index]) if world > 1 else model
    parameters = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(parameters, lr=args.base_lr, weight_decay=args.wd)
    criterion = nn.CrossEntropyLoss(ignore_index=-100)

    n_local = args.num_samples // world
    derm_all, clinic_all = synthetic_split(n_local, args.img_sz, dev, args.seed + local_rank)
    index_all = torch.arange(local_rank * n_local, (local_rank + 1) * n_local, device=dev)
    n_batches = n_local // bs  # drop_last, as the reference's train loader

    def loader(perm=None):
        order = perm if perm is not None else torch.arange(n_local, device=dev)
        for i in range(n_batches):
            sel = order[i * bs:(i + 1) * bs]
            yield index_all[sel], derm_all[sel], clinic_all[sel]

    model.eval() if not args.finetune_backbone else model.train()
    local_memory_index, local_memory_embeddings = init_memory(loader, model)
    gk = torch.Generator().manual_seed(args.seed)
    history = []
    os.makedirs(args.log_path, exist_ok=True)
    for epoch in range(args.epochs):
        t0 = time.time()
        all_assignments = [cluster_memory(args, proto, proto.weight.size(0), local_memory_index,
                                          local_memory_embeddings[i % len(local_memory_embeddings)], generator=gk)
                           for i, proto in enumerate(model.prototypes)]
        if getattr(args, "probe", None) is not None:  # tests: what every rank ended up with after the broadcast
            args.probe["assignments"] = [a.cpu() for a in all_assignments]
            args.probe["prototypes"] = [p.weight.detach().cpu().clone() for p in model.prototypes]
        if args.finetune_backbone:
            model.train()
        else:  # mlc_train.py:230-235
            model.extractor.eval()
            model.projectors.train()
            model.mlc_sa.train()
            model.prototypes.train()
        perm = torch.randperm(n_local, device=dev)
        start_idx, total, seen = 0, 0.0, 0
        for it, (idx, derm, clinic) in enumerate(loader(perm)):
            proj_feats, preds = wrapped(derm, clinic)
            loss = 0
            for pred, assignment in zip(preds, all_assignments):
                loss = loss + criterion(pred / args.temperature, assignment[idx])
            loss = loss / len(all_assignments)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            nb = idx.shape[0]
            local_memory_index[start_idx:start_idx + nb] = idx
            for i in range(len(local_memory_embeddings)):
                local_memory_embeddings[i][start_idx:start_idx + nb] = proj_feats[i].detach()
            start_idx += nb
            total += float(loss.detach()) * nb
            seen += nb
            if local_rank == 0 and it % args.print_freq == 0:
                print(f"Train epoch: [{epoch}][{it}/{n_batches}] Loss {float(loss.detach()):.4f}", flush=True)
        history.append(total / max(seen, 1))
        if getattr(args, "probe", None) is not None:
            args.probe["model"] = model
        if local_rank == 0:
            print(f"epoch {epoch}: loss {history[-1]:.4f}, {time.time() - t0:.1f} s", flush=True)
            state = {"epoch": epoch + 1, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
            torch.save(state, os.path.join(args.log_path, "checkpoint.pth.tar"))
            if (epoch + 1) % args.save_freq == 0 or (epoch + 1) == args.epochs:
                torch.save({"epoch": epoch + 1, "state_dict": model.state_dict()},
                           os.path.join(args.log_path, f"ckp_{epoch}.pth"))
    if world > 1:
        dist.destroy_process_group()
    return history


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
