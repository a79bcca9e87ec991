"""Data-parallel path on CPU: world_size-2 gloo.

1. Numerics of the design, with the CPU oracle as the compute: SyncBatchNorm by summed (sum x, sum x^2, count)
   plus rank-averaged gradients equals a single process that normalises with full-batch statistics and
   averages the per-shard (local-negatives) NT-Xent losses -- the reference's DDP + SyncBatchNorm semantics
   (tools/backbone_train.py:510,522).
2. Control flow of the product's trainer with the C ABI stubbed (tests/fakelib.py): both ranks issue the same
   collective sequence (230 forward + 230 backward statistic all-reduces, then the gradient buckets) and the
   fused AdamW receives grad_scale = 1/world.
"""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(fn, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, ok, payload = q.get(timeout=600)
        results[r] = (ok, payload)
    for p in procs:
        p.join(timeout=60)
    for r, (ok, payload) in sorted(results.items()):
        assert ok, f"rank {r} failed:\n{payload}"
    return {r: payload for r, (ok, payload) in results.items()}


def _entry(fn, rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(4 if world <= 2 else 1)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        out = fn(rank, world)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, True, out))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        dist.all_reduce(y)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g)
        return g


def _oracle_rank(rank, world):
    from oracle import procedural, sm3_oracle as O
    Bl, size, seed, T = (3 if world <= 2 else 2), 32, 5, 0.1
    Bg = Bl * world
    state = procedural.make_state_dict(seed=seed)
    derm_np, clinic_np = procedural.make_pair_batch(Bg, size, seed)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    dt = torch.float64
    P, B = O.split_state(state, dt)
    derm = [torch.from_numpy(a[sl]).to(dt) for a in derm_np]
    clinic = [torch.from_numpy(a[sl]).to(dt) for a in clinic_np]
    outs = O.sm3_v32_forward(P, B, derm, clinic, 0, T, True, stat_reduce=_AllReduceSum.apply)
    loss = O.sm3_loss(outs, 0)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in P.values()])
    dist.all_reduce(flat)
    flat /= world  # DDP: mean over ranks
    result = None
    if rank == 0:
        # single-process reference: full-batch BN statistics, per-shard local-negative losses, averaged
        torch.set_num_threads(8)  # (the other ranks wait in the all-gather below)
        P2, B2 = O.split_state(state, dt)
        dfull = [torch.from_numpy(a).to(dt) for a in derm_np]
        cfull = [torch.from_numpy(a).to(dt) for a in clinic_np]
        zs = O.sm3_v32_projections(P2, B2, dfull, cfull, 0, True)
        total = 0.0
        for r in range(world):
            rows = torch.cat([torch.arange(r * Bl, (r + 1) * Bl), Bg + torch.arange(r * Bl, (r + 1) * Bl)])
            terms = [O.cross_entropy_zero_label(O.ntxent_logits(z[rows], T)[0]) for z in zs]
            total = total + (terms[0] + terms[1] + 0.5 * terms[2] + 0.5 * terms[3]) / world
        total.backward()
        ref = torch.cat([p.grad.reshape(-1) for p in P2.values()])
        bkeys = [k for k in B if k.endswith(("running_mean", "running_var"))]
        buf_err = max(float((B[k] - B2[k]).abs().max()) for k in bkeys)
        result = {"grad_rel": float((flat - ref).norm() / ref.norm()), "buf_err": buf_err,
                  "nbt": int(B["derm_backbone.encoder.bn1.num_batches_tracked"])}
    losses = [torch.zeros(1, dtype=dt) for _ in range(world)]
    dist.all_gather(losses, loss.detach().reshape(1))
    if rank == 0:
        result["loss_mean"] = float(sum(losses) / world)
        result["loss_ref"] = float(total)
    return result


def test_syncbn_and_grad_averaging_semantics_gloo():
    out = _spawn(_oracle_rank)[0]
    assert out["grad_rel"] < 1e-9, out
    assert out["buf_err"] < 1e-10, out
    assert abs(out["loss_mean"] - out["loss_ref"]) < 1e-10, out
    assert out["nbt"] == 2


def test_syncbn_and_grad_averaging_semantics_gloo_world8():
    """The same at BASELINE config 3's rank count: 8 ranks x 2 pairs (a rank's NT-Xent rows see their positive
    and the two rows of the other local pair), statistics and gradients summed over 8."""
    out = _spawn(_oracle_rank, world=8)[0]
    assert out["grad_rel"] < 1e-9, out
    assert out["buf_err"] < 1e-10, out
    assert abs(out["loss_mean"] - out["loss_ref"]) < 1e-10, out
    assert out["nbt"] == 2


def _trainer_rank(rank, world):
    from fakelib import installed
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(0)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    log = []
    real_all_reduce = dist.all_reduce

    def spy(t, *a, **k):
        log.append((tuple(t.shape), str(t.dtype), bool(k.get("async_op", False))))
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = spy
    try:
        with installed() as fake:
            tr = SM3Trainer(model, lr=1e-3)
            assert tr.dp and tr.world == world
            x = [torch.randn(2, 3, 32, 32) for _ in range(4)]
            tr.step(x[:2], x[2:])
            st = tr._engine().store
            grad_elems = sum(s[0] for s, dt, a in [(l[0], l[1], l[2]) for l in log] if dt == "torch.float32")
            total = st.total
    finally:
        dist.all_reduce = real_all_reduce
    stats = [l for l in log if l[1] == "torch.float64"]
    buckets = [l for l in log if l[1] == "torch.float32"]
    return {"n_stats": len(stats), "n_buckets": len(buckets), "bucket_elems": grad_elems, "total": total,
            "seq": [(l[0], l[1]) for l in log], "async": all(l[2] for l in buckets)}


def test_trainer_collective_pattern_gloo():
    res = _spawn(_trainer_rank)
    a, b = res[0], res[1]
    assert a["seq"] == b["seq"]                       # identical collective order on both ranks: no deadlock by construction
    # 214 forward + 214 backward SyncBN reductions (230 + 230 BatchNorm evaluations, per-view passes at this
    # batch size): the two BatchNorms that meet at a downsample block's join share one all-reduce in each direction
    # (engine.conv_bn `pending`, engine.bn_backward_join), 4 blocks x 4 encoder passes fewer than one per BatchNorm
    assert a["n_stats"] == 428
    assert a["bucket_elems"] == a["total"]            # gradient buckets tile the flat buffer exactly once
    assert 4 <= a["n_buckets"] <= 16 and a["async"]


def test_trainer_collective_pattern_gloo_world8():
    """BASELINE config 3's rank count (8 x MI355X) on the CPU: the product trainer's collective sequence -- per-lane
    SyncBatchNorm statistic all-reduces, gradient buckets as they become final, AdamW with grad_scale = 1/8 -- is the same
    list on all eight ranks (no deadlock by construction), its counts do not depend on the world size, and the buckets tile
    the flat gradient buffer exactly once.  No hardware with 8 GPUs was available to any round: config 3 stays unmeasured."""
    res = _spawn(_trainer_rank, world=8)
    assert len(res) == 8
    for r in range(1, 8):
        assert res[r]["seq"] == res[0]["seq"], r
    a = res[0]
    assert a["n_stats"] == 428
    assert a["bucket_elems"] == a["total"]
    assert 4 <= a["n_buckets"] <= 16 and a["async"]


def _kmeans_cpu(emb, K, iters=10, generator=None):
    """CPU restatement of the reference's spherical k-means (tools/mlc_train.py:146-177) with sm3hip.mlc.spherical_kmeans'
    interface: the HIP kernels are checked against this same restatement in tests/test_mlc.py (GPU)."""
    N = emb.shape[0]
    c = emb[torch.randperm(N, generator=generator)[:K]].double()
    e = emb.double()
    for it in range(iters + 1):
        a = (e @ c.t()).max(1).indices
        if it == iters:
            break
        for k in range(K):
            if (a == k).any():
                c[k] = e[a == k].sum(0) / (a == k).sum()
        c = torch.nn.functional.normalize(c, dim=1)
    return c.float(), a


def test_kmeans_restatement_reproduces_the_reference_function(golden_dir):
    """The CPU restatement above against the REFERENCE'S OWN `cluster_memory` (tools/mlc_train.py:116-189, run on CPU by
    oracle/gen_kmeans_golden.py, tests/golden/mlc_kmeans_ref.npz): same initial centroids from the same seed, same assignment
    of every memory index, same centroids -- so the restatement the gloo and GPU tests lean on is pinned, not asserted."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, "mlc_kmeans_ref.npz"))
    ncases = len([k for k in g.files if k.endswith("_meta")])
    assert ncases == 4
    for ci in range(ncases):
        N, D, K, kseed = [int(v) for v in g[f"c{ci}_meta"]]
        emb, index = torch.from_numpy(g[f"c{ci}_emb"]), torch.from_numpy(g[f"c{ci}_index"])
        cent, a = _kmeans_cpu(emb, K, generator=torch.Generator().manual_seed(kseed))
        assign = torch.full((N,), -100, dtype=torch.long)
        assign[index] = a                                   # mlc_train.py:181: logged by memory index
        assert torch.equal(assign, torch.from_numpy(g[f"c{ci}_assign"])), ci
        assert float((cent - torch.from_numpy(g[f"c{ci}_centroids"])).abs().max()) < 2e-6, ci


def _cluster_rank(rank, world):
    """tools/mlc_train.py:136-143,185-186 under data parallelism: every rank holds a shard of the memory bank; rank 0
    gathers it, clusters, and broadcasts centroids + assignments."""
    import importlib.util
    import types
    from sm3hip import mlc
    spec = importlib.util.spec_from_file_location("sm3_mlc_train", os.path.join(ROOT, "skin-sm3_amd", "tools", "mlc_train.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    mlc.spherical_kmeans = _kmeans_cpu                      # the clustering itself is a HIP kernel: GPU-tested elsewhere
    N_local, D, K = 37, 32, 5
    g = torch.Generator().manual_seed(100)                   # the whole bank, identical on both ranks; each keeps its shard
    centers = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
    full = torch.nn.functional.normalize(centers[torch.randint(0, K, (world * N_local,), generator=g)]
                                         + 0.2 * torch.randn(world * N_local, D, generator=g), dim=1)
    perm = torch.randperm(world * N_local, generator=g)      # DistributedSampler-style interleaved sample indices
    local_index = perm[rank::world].contiguous()
    local_emb = full[local_index].contiguous()
    proto = torch.nn.Linear(D, K, bias=False)
    args = types.SimpleNamespace(world_size=world, rank=rank)
    gk = torch.Generator().manual_seed(7)                    # only rank 0 draws from it
    assign = mt.cluster_memory(args, proto, K, local_index, local_emb, generator=gk)
    # single-process answer on the bank in the order rank 0 assembles it (rank-major)
    order = torch.cat([perm[r::world] for r in range(world)])
    c_ref, a_ref = _kmeans_cpu(full[order], K, generator=torch.Generator().manual_seed(7))
    want = torch.empty(world * N_local, dtype=torch.long)
    want[order] = a_ref
    # plain lists: tensors handed through a multiprocessing queue need their producer alive
    return {"assign": assign.tolist(), "weight": proto.weight.detach().tolist(), "want": want.tolist(),
            "c_ref": c_ref.tolist()}


def test_mlc_cluster_memory_gather_kmeans_broadcast_gloo():
    res = _spawn(_cluster_rank)
    a, b = res[0], res[1]
    assert a["assign"] == b["assign"] and a["weight"] == b["weight"]   # both ranks hold rank 0's result
    assert a["assign"] == a["want"]                          # every sample's cluster, addressed by its dataset index
    assert min(a["assign"]) >= 0                             # no -100 left: the shards cover the bank
    assert a["weight"] == a["c_ref"]                         # the centroids became the prototype weights


def test_mlc_cluster_memory_gather_kmeans_broadcast_gloo_world8():
    """BASELINE config 4 is an 8-GPU configuration: eight shards of the memory bank -> rank 0 -> k-means -> broadcast."""
    res = _spawn(_cluster_rank, world=8)
    a = res[0]
    for r in range(1, 8):
        assert res[r]["assign"] == a["assign"] and res[r]["weight"] == a["weight"], r
    assert a["assign"] == a["want"] and min(a["assign"]) >= 0 and len(a["assign"]) == 8 * 37
    assert a["weight"] == a["c_ref"]


def test_p2p_mailbox_layout_has_room_for_eight_ranks():
    """csrc/p2p.hip addresses a mailbox as data[slot][source rank][element] + flag[slot][source rank][block].  Checked on
    the host, from the library's own struct (sm3_p2p_layout): with 8 source ranks x 2 slots every data area and every flag
    is inside sm3_p2p_mailbox_bytes(), no two overlap, the blocks of the largest exchange have a flag each, and the largest
    message the engine exchanges -- bn3 and the downsample BatchNorm of layer 4 in one collective, two views, sum and sum of
    squares of 2 048 channels each: 2 x 2 x 2 x 2 048 doubles -- fits sm3_p2p_max_elems()."""
    import ctypes as C
    from sm3hip import _lib
    lib = _lib.load()
    W = lib.sm3_p2p_max_world()
    assert W == 8
    total, nmax = lib.sm3_p2p_mailbox_bytes(), lib.sm3_p2p_max_elems()
    assert nmax >= 2 * 2 * 2 * 2048
    spans = []
    per_block = None
    d0, db, fo, epb = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    for slot in (0, 1):
        for src in range(W):
            block = 0
            while lib.sm3_p2p_layout(slot, src, block, C.byref(d0), C.byref(db), C.byref(fo), C.byref(epb)) == 0:
                per_block = epb.value
                if block == 0:
                    assert db.value >= nmax * 8
                    spans.append((d0.value, d0.value + db.value, ("data", slot, src)))
                spans.append((fo.value, fo.value + 8, ("flag", slot, src, block)))
                block += 1
            assert block * per_block >= nmax, (block, per_block)       # every element of the largest message has a block
    assert lib.sm3_p2p_layout(0, W, 0, C.byref(d0), C.byref(db), C.byref(fo), None) == -1   # a ninth rank is refused
    assert lib.sm3_p2p_layout(2, 0, 0, C.byref(d0), C.byref(db), C.byref(fo), None) == -1   # ... and a third slot
    spans.sort()
    assert spans[0][0] >= 0 and spans[-1][1] <= total
    for (a0, a1, wa), (b0, b1, wb) in zip(spans, spans[1:]):
        assert a1 <= b0, (wa, wb)
    assert all(s[0] % 8 == 0 for s in spans)                           # 64-bit atomics need 8-byte alignment
