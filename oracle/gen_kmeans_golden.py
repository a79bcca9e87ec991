"""TEST INFRASTRUCTURE ONLY (build container): golden vectors for the spherical k-means of the multi-label clustering tool,
produced by the REFERENCE'S OWN FUNCTION.

`tools/mlc_train.py:116-195` (`cluster_memory`, `get_indices_sparse`) cannot be imported as a module here -- the file's imports
pull in cv2 and torchvision.transforms, which this image lacks -- so the two function definitions are taken out of the
reference's source with `ast` AT GENERATION TIME (nothing of them is stored in this repository), compiled as they are and run
on CPU: `Tensor.cuda` is the identity for the duration, `torch.distributed` is a one-rank gloo group (the function gathers and
broadcasts).  Inputs are seeded synthetic memory banks of derm7pt's sizes; the stored outputs are what the reference returns:
the assignment of every memory index and the centroids it copies into the prototype layer.

    python oracle/gen_kmeans_golden.py            # writes tests/golden/mlc_kmeans_ref.npz
"""
import ast
import os
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from scipy.sparse import csr_matrix

REF = "/root/reference/tools/mlc_train.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_functions():
    src = open(REF).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("cluster_memory", "get_indices_sparse")]
    assert len(keep) == 2
    ns = {"torch": torch, "nn": nn, "dist": dist, "np": np, "csr_matrix": csr_matrix}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["cluster_memory"]


def bank(N, D, K, noise, seed):
    g = torch.Generator().manual_seed(seed)
    centers = nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    emb = nn.functional.normalize(centers[torch.randint(0, K, (N,), generator=g)] + noise * torch.randn(N, D, generator=g), dim=1)
    index = torch.randperm(N, generator=g)          # the order in which the loader visited the samples (mlc_train.py:93-112)
    return emb.contiguous(), index


def main():
    cluster_memory = reference_functions()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", world_size=1, rank=0)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        # (N, D, K, noise, data seed, k-means seed): derm7pt's 413 training cases, label embeddings of 512 (run.sh) down to 128 dimensions, the label
        # cardinalities of NUM_CLASSES (mlc_train.py:55); one bank whose clusters overlap, one with fewer samples
        cases = [(413, 512, 5, 0.3, 4, 7), (413, 256, 3, 0.6, 5, 11), (413, 128, 2, 0.9, 6, 3), (96, 128, 3, 0.4, 8, 21)]
        for ci, (N, D, K, noise, dseed, kseed) in enumerate(cases):
            emb, index = bank(N, D, K, noise, dseed)
            proto = nn.Linear(D, K, bias=False)
            args = types.SimpleNamespace(world_size=1, rank=0)
            torch.manual_seed(kseed)                 # the reference draws its initial centroids from the global generator
            assign = cluster_memory(args, proto, K, index.clone(), emb.clone())
            out[f"c{ci}_emb"] = emb.numpy()
            out[f"c{ci}_index"] = index.numpy()
            out[f"c{ci}_meta"] = np.array([N, D, K, kseed], dtype=np.int64)
            out[f"c{ci}_assign"] = assign.numpy()
            out[f"c{ci}_centroids"] = proto.weight.detach().numpy().copy()
            print(f"case {ci}: N={N} D={D} K={K} cluster sizes {np.bincount(assign.numpy(), minlength=K).tolist()}")
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    path = os.path.join(ROOT, "tests", "golden", "mlc_kmeans_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
