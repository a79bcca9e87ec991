"""csrc/p2p.hip: the opt-in peer-to-peer SyncBatchNorm statistics exchange.  Two PROCESSES share the test box's
one GPU and map each other's mailboxes through hipIpc -- every line of the kernel and of sm3hip/p2p.py runs; the xGMI hop
between devices is what a one-GPU box cannot exercise."""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(4)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)


def _exchange_main(rank, world, port, q):
    try:
        _setup(rank, world, port)
        from sm3hip import ops
        from sm3hip.p2p import P2PStatSync
        dev = torch.device("cuda:0")
        sync = P2PStatSync(["a", "b"], dev, timeout_s=20.0)
        streams = {"a": torch.cuda.Stream(), "b": torch.cuda.Stream()}
        sizes = [256, 8192, 16384, 100, 4096, 1, 1025, sync.max_elems]
        bad = 0
        outs = {"a": [], "b": []}
        for i in range(120):
            for lane in ("a", "b"):
                n = sizes[(i + (lane == "b")) % len(sizes)]
                with torch.cuda.stream(streams[lane]), ops.stream_scope():
                    g = torch.Generator(device=dev).manual_seed(1000 * i + (7 if lane == "b" else 0) + rank)
                    t = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
                    sync(lane, t)
                    outs[lane].append((i, n, t))
        torch.cuda.synchronize()
        sync.check()
        for lane in ("a", "b"):
            for i, n, t in outs[lane]:
                ref = torch.zeros(n, dtype=torch.float64, device=dev)
                for r in range(world):  # rank order, as the kernel adds them
                    g = torch.Generator(device=dev).manual_seed(1000 * i + (7 if lane == "b" else 0) + r)
                    ref += torch.randn(n, dtype=torch.float64, device=dev, generator=g)
                bad += int(not torch.equal(t, ref))
        # a peer that never shows up: the kernel gives up after its timeout instead of spinning forever
        late = P2PStatSync(["x"], dev, timeout_s=0.5)
        after = {"nan": True, "ms": 0.0, "poll": True}
        if rank == 0:
            t = torch.ones(64, dtype=torch.float64, device=dev)
            late("x", t)
            torch.cuda.synchronize()
            timed_out = bool(torch.isnan(t).all())  # the failed exchange poisons its sums
            try:
                late.check()
                timed_out = False
            except RuntimeError:
                pass
            # once failed, always failed and never waiting again: 100 more exchanges of the largest size return NaN at once
            import time
            ts = [torch.ones(late.max_elems, dtype=torch.float64, device=dev) for _ in range(100)]
            t0 = time.perf_counter()
            for u in ts:
                late("x", u)
            torch.cuda.synchronize()
            after["ms"] = (time.perf_counter() - t0) * 1e3
            after["nan"] = all(bool(torch.isnan(u).all()) for u in ts)
            try:  # the non-blocking form a trainer uses: the second call sees what the first one copied back
                late.poll()
                torch.cuda.synchronize()
                late.poll()
                after["poll"] = False
            except RuntimeError:
                pass
        else:
            timed_out = True
        dist.barrier()
        late.close()
        sync.close()
        dist.destroy_process_group()
        q.put((rank, True, {"bad": bad, "timed_out": timed_out, "after": after, "kind": sync.memory_kind}))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


def _spawn(fn, world=2, extra=()):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, ok, payload = q.get(timeout=180)
        assert ok, f"rank {r} failed:\n{payload}"
        res[r] = payload
    for p in procs:
        p.join(timeout=60)
    return res


def test_two_processes_exchange_sums_through_ipc_mailboxes():
    """240 exchanges on two lanes (two streams, two mailbox sets) with sizes from 1 to the maximum: every result equals the
    rank-ordered fp64 sum bit for bit on both ranks; an exchange whose peer never arrives ends by its timeout."""
    res = _spawn(_exchange_main)
    for r in (0, 1):
        assert res[r]["bad"] == 0 and res[r]["timed_out"], res[r]
        assert res[r]["kind"] in ("finegrained", "coarse"), res[r]
    print("mailbox memory:", res[0]["kind"], "| 100 exchanges after a timeout:", round(res[0]["after"]["ms"], 2), "ms")
    assert res[0]["after"]["nan"] and res[0]["after"]["poll"], res[0]
    assert res[0]["after"]["ms"] < 10.0, res[0]  # (one wait of the 0.5 s timeout alone would be 500 ms)


def _missing_peer_main(rank, world, port, q):
    try:
        os.environ["SM3_SYNCBN_P2P"] = "1"
        os.environ["SM3_P2P_TIMEOUT_S"] = "0.5"
        _setup(rank, world, port)
        from oracle import procedural
        from sm3hip.trainer import SM3Trainer
        from src.models.simclr import SimCLRSkinV32
        dev = torch.device("cuda:0")
        state = procedural.make_state_dict(seed=21)
        derm_np, clinic_np = procedural.make_pair_batch(16, 64, 21)
        model = SimCLRSkinV32("resnet50", None, 128, 0.1)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        model.sm3_dtype = torch.bfloat16
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(dev)
        tr = SM3Trainer(model, lr=1e-3, target_momentum=0.99)
        tr._engine()  # both ranks map each other's mailboxes ...
        tr._bucket_ready = lambda *a: None  # (no gradient all-reduce: this is about the statistics exchange, and rank 1 is absent)
        out = {"nan": True, "check": True, "step": True, "s": 0.0, "params_kept": True, "count_kept": True}
        if rank == 0:  # ... but only rank 0 steps: its peer never raises a flag
            import time
            batch = ([torch.from_numpy(a).to(dev) for a in derm_np], [torch.from_numpy(a).to(dev) for a in clinic_np])
            tr._engine().prepare(dev)
            before = tr._engine().store.flat_p.clone()
            t0 = time.perf_counter()
            loss = float(tr.step(*batch))
            out["s"] = time.perf_counter() - t0
            out["nan"] = loss != loss
            # the poisoned gradients were NOT applied: parameters bit-identical, AdamW moments still zero
            st = tr._engine().store
            out["params_kept"] = bool(torch.equal(st.flat_p, before)) and float(tr.m.abs().max()) == 0.0 \
                and float(tr.v.abs().max()) == 0.0 and not bool(torch.isfinite(st.flat_g).all())
            # ... and the momentum target did not move: it started as a copy of the online parameters
            out["params_kept"] = out["params_kept"] and bool(torch.equal(tr.flat_target, before))
            try:
                tr.check()
                out["check"] = False
            except RuntimeError:
                pass
            # the step counter a checkpoint would carry is that of the last APPLIED step: none
            out["count_kept"] = tr.step_count == 0 and tr.steps_taken() == 0 and \
                (tr.optimizer_state_dict()["state"][0]["step"].item() == 0.0)
            try:
                tr.step(*batch)
                out["step"] = False
            except RuntimeError:
                pass
            out["count_kept"] = out["count_kept"] and tr.step_count == 0
        dist.barrier()
        tr.close(barrier=False)
        dist.destroy_process_group()
        q.put((rank, True, out))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


def test_trainer_step_with_a_missing_peer_fails_loudly():
    """SM3_SYNCBN_P2P=1, rank 1 never steps: rank 0's first exchange runs into its timeout ONCE (0.5 s here), every later one
    returns at once, the step's loss is NaN, SM3Trainer.check() raises and so does the next step() -- training cannot carry
    on with un-reduced statistics -- and the optimizer update of the poisoned step is skipped on the device, so
    a caller that catches the error still holds the model of the last good step."""
    res = _spawn(_missing_peer_main)
    assert res[0]["nan"] and res[0]["check"] and res[0]["step"], res[0]
    assert res[0]["params_kept"], res[0]
    assert res[0]["count_kept"], res[0]
    assert res[0]["s"] < 15.0, res[0]  # 220 exchanges x 0.5 s would be 110 s


def _step_main(rank, world, port, q, p2p):
    try:
        if p2p:
            os.environ["SM3_SYNCBN_P2P"] = "1"
        _setup(rank, world, port)
        from oracle import procedural
        from sm3hip.trainer import SM3Trainer
        from src.models.simclr import SimCLRSkinV32
        dev = torch.device("cuda:0")
        Bl = 32
        state = procedural.make_state_dict(seed=21)
        derm_np, clinic_np = procedural.make_pair_batch(Bl * world, 64, 21)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        out = {}
        for dt in (torch.float32, torch.bfloat16):
            model = SimCLRSkinV32("resnet50", None, 128, 0.1)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
            model.sm3_dtype = dt
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(dev)
            tr = SM3Trainer(model, lr=1e-3)
            assert tr.dp and tr.sync_bn
            loss = float(tr.step([torch.from_numpy(a[sl]).to(dev) for a in derm_np], [torch.from_numpy(a[sl]).to(dev) for a in clinic_np]))
            torch.cuda.synchronize()
            if p2p:
                tr._p2p.check()
            sd = model.state_dict()
            out[str(dt)] = (loss, float(tr._engine().store.flat_g.double().norm()),
                            float(sd["derm_backbone.encoder.layer4.2.bn3.running_var"].double().sum()),
                            float(sd["clinic_backbone.encoder.layer1.0.downsample.1.running_mean"].double().sum()))
            if p2p:
                tr._p2p.close()
            del tr, model
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, True, out))
    except Exception:
        q.put((rank, False, traceback.format_exc()))


def test_dp_step_with_p2p_statistics_equals_the_all_reduce_path():
    """Two ranks, real kernels, both views in one batch: the step with SM3_SYNCBN_P2P=1 against the same step with the
    torch.distributed all-reduce (gloo here).  Both add the two ranks' fp64 sums in rank order, so every statistic -- and with
    it the loss and the running buffers -- is the same number."""
    a = _spawn(_step_main, extra=(True,))
    b = _spawn(_step_main, extra=(False,))
    for r in (0, 1):
        for dt in a[r]:
            pa, pb = a[r][dt], b[r][dt]
            print(r, dt, pa, pb)
            tol = 1e-6 if "float32" in dt else 1e-3  # weight gradients accumulate with float atomics (order varies run to run)
            assert abs(pa[0] - pb[0]) < tol * abs(pb[0]), (dt, pa, pb)
            assert abs(pa[1] - pb[1]) < max(tol, 1e-4) * abs(pb[1]), (dt, pa, pb)
            assert abs(pa[2] - pb[2]) < tol * abs(pb[2]) and abs(pa[3] - pb[3]) < tol * abs(pb[3]) + 1e-6, (dt, pa, pb)
