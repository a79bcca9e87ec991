#!/usr/bin/env python
"""SM3 pre-training throughput on MI355X: `python bench.py --gpus N --steps K --warmup W`.

One "step" = one optimizer step of SimCLRSkinV32(resnet50) on a batch of synthetic pairs per GPU
(BASELINE.json configs[1]: 224x224 derm+clinical pairs, ResNet-50 x2, batch 256 per GPU, bf16 MFMA with fp32
accumulate): forward (4 encoder passes + 6 projector passes), 4-term NT-Xent loss, backward, (N>1: RCCL
gradient all-reduce + SyncBN statistics), fused AdamW.  Inputs are resident in HBM before the timed region.
For N>1 the driver launches this file under torch.distributed.run, one rank per GPU (weak scaling).

Prints ONE JSON line on rank 0 (see the contract in the task description) with two extra objects:
  roofline      achieved TFLOP/s of the dominant kernel (the 128x128 bf16 gather-GEMM convolution,
                conv_igemm_kernel<bf16_t,128,128,2,2>): algorithmic FLOPs of its launches / their summed
                duration, timed live with HIP events on the launch stream during the timed steps
  cpu_baseline  the CPU oracle (oracle/sm3_oracle.py, kind "port") timed on this box's host cores on a bounded
                sample of the same workload (B=8 pairs per step, fp32)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "skin-sm3_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Kernel arguments in device memory instead of host memory (a ROCm runtime switch; the HIP runtime reads it when it
# initialises, so it is set before `import torch`): a kernel that starts no longer fetches its argument block over PCIe.
# With ~1 160 dependent launches per step that is +2.2 % of the two-lane step and +3.5 % single-lane
# (profiles/r06_dev_kernarg_ab.txt).  An explicit HIP_FORCE_DEV_KERNARG in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_BYTES = 8.0e12  # MI355X_MICROARCH.md: HBM3E spec peak
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense, MI355X_MICROARCH.md "Chip-level parameters"
FLOP_PER_PAIR_224 = 98.5e9


def kernel_source_sha():
    """sha256 over the sources of the dominant kernel: ties a committed PMC measurement to the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_igemm.hip", "conv_common.h", "common.h"):
        h.update(open(os.path.join(ROOT, "skin-sm3_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def replayed_traffic(dtype, B, S):
    """HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs of this same command, FETCH_SIZE doubled per the gfx950 correction; scratch/collect_traffic.py).
    Counters cannot be read from inside the process, so the newest committed measurement is REPLAYED -- only for the
    workload it was taken on and only while the kernel sources still hash to what it recorded; otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_conv_igemm_b256_bf16.json")))
    if not (dtype == "bf16" and B == 256 and S == 224 and files):
        return None, "none for this workload"
    rec = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if rec.get("kernel_source_sha") != kernel_source_sha():
        return None, f"{rel} is stale (kernel sources changed since it was collected): not reported"
    return round(rec["hbm_bytes_per_launch"]), f"{rel} (replayed; rocprofv3 --pmc passes of this command, not measured in this run)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="bf16 = BASELINE.json configs[1] (default); f16 = the reference's AMP type, with dynamic loss scaling "
                         "(configs[4]); f32 = exact-f32 MFMA parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-dtypes", action="store_true",
                    help="skip the two short extra loops that time the same step in the other arithmetic modes "
                         "(config.other_modes; N=1 default run only)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--workload", default="pretrain", choices=["pretrain", "linear_probe", "inference", "mlc_train"],
                    help="pretrain = BASELINE.json's metric (default); linear_probe = SURVEY.md 8f-1 (tools/backbone_eval.py "
                         "--finetune fc step at run.sh's batch 128: frozen eval-mode encoders + 8 trained heads)")
    ap.add_argument("--global-negatives", action="store_true",
                    help="opt-in north_star mode: NT-Xent against the all-gathered projections of every rank (not the "
                         "reference's local negatives; changes the loss)")
    ap.add_argument("--metadata-dim", type=int, default=0,
                    help="opt-in north_star extension: synthetic N(0,1) metadata of this width through the metadata-MLP branch "
                         "(two more NT-Xent terms); 0 = the reference's model")
    ap.add_argument("--target-momentum", type=float, default=None,
                    help="opt-in north_star extension: momentum-updated target encoders (one more forward pass per step)")
    ap.add_argument("--single-lane", action="store_true",
                    help="run the derm and clinic branches on ONE stream (diagnostic: per-kernel durations without the "
                         "other lane's kernels sharing the chip -- what roofline.achieved is measured on)")
    ap.add_argument("--breakdown", default=None, help="write a per-kernel-class time/FLOP/byte table (one extra, "
                                                       "untimed, fully instrumented step) to this file")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU box gives
    one GPU's share of the host; spinning up one thread per visible core oversubscribes it badly)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, q // p))
        except Exception:
            pass
    return min(cores, int(os.environ.get("SM3_CPU_BASELINE_THREADS", "16")))


def cpu_baseline(batch, img, steps):
    """Times the CPU oracle's full training step (forward, 4-term loss, backward, AdamW) on the host cores."""
    from oracle import procedural, sm3_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} threads, B={batch}", file=sys.stderr, flush=True)
    P, B = O.split_state(procedural.make_state_dict(seed=0), torch.float32)
    g = torch.Generator().manual_seed(3407)
    derm = [torch.randn(batch, 3, img, img, generator=g) for _ in range(2)]
    clinic = [torch.randn(batch, 3, img, img, generator=g) for _ in range(2)]
    opt = {}
    O.train_step(P, B, derm, clinic, 0, 0.1, opt, lr=1e-6)  # warm-up
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        O.train_step(P, B, derm, clinic, 0, 0.1, opt, lr=1e-6)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {times[-1]:.2f} s", file=sys.stderr, flush=True)
    med = sorted(times)[len(times) // 2]
    return {"value": batch / med, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle/sm3_oracle.py train_step, B={batch} pairs/step, {img}x{img}, fp32, 1 warm-up + "
                      f"{steps} timed steps, median {med:.2f} s/step"}


def linear_probe_bench(args):
    """Secondary line: the linear-probe step of tools/backbone_eval.py:98-112 (--finetune fc), single GPU."""
    from sm3hip import ops, profiler
    from src.models.baseline import Baseline, NUM_CLASSES
    dev = torch.device("cuda", 0)
    torch.manual_seed(3407)
    B, S = (128 if args.batch == 256 else args.batch), args.img
    m = Baseline("resnet50", None)
    m.freeze_backbone()
    for bb in (m.derm_backbone, m.clinic_backbone):
        bb.sm3_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    m.to(dev).eval()
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-2)
    crit = torch.nn.CrossEntropyLoss()
    g = torch.Generator(device=dev).manual_seed(3407)
    derm = torch.randn(B, 3, S, S, device=dev, generator=g)
    clinic = torch.randn(B, 3, S, S, device=dev, generator=g)
    labels = torch.stack([torch.randint(0, n, (B,), device=dev, generator=g) for n in NUM_CLASSES], dim=1)

    def step():
        outs = m([derm, clinic])
        loss = sum(crit(o, labels[:, i]) for i, o in enumerate(outs)) / 8
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    prof = profiler.Profiler(only={"conv_gemm_128x128"})
    torch.cuda.synchronize()
    ops.set_profiler(prof)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)
    dom = prof.summary().get("conv_gemm_128x128", {"flops": 0.0, "ms": 0.0, "launches": 0})
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    print(json.dumps({
        "metric": "SM3 linear-probe pairs/sec (224x224, frozen ResNet-50 x2 + 8 heads)", "value": round(B * args.steps / elapsed, 2),
        "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"tools/backbone_eval.py --finetune fc step, Baseline(resnet50 x2), batch {B}, {S}x{S}",
                   "global_batch": B, "parallelism": "dp1", "loss": round(float(loss.detach()), 5)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None,
                     "kernel": "conv_igemm_kernel<bf16_t,128,128,2,2,*> with the conv+evalBN+ReLU epilogue"}}), flush=True)


def inference_bench(args):
    """Secondary line: forward of the reference's inference.py model (eval mode), single GPU -- HIP encoders with the
    fused conv+BN epilogue, label projectors / 8-token transformer layer / prototype heads on csrc/heads.hip."""
    from sm3hip import ops, profiler
    import inference
    dev = torch.device("cuda", 0)
    torch.manual_seed(3407)
    B, S = (128 if args.batch == 256 else args.batch), args.img
    m = inference.build_model()
    for bb in (m.extractor.derm_backbone, m.extractor.clinic_backbone):
        bb.sm3_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    m.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(3407)
    derm = torch.randn(B, 3, S, S, device=dev, generator=g)
    clinic = torch.randn(B, 3, S, S, device=dev, generator=g)
    with torch.no_grad():
        for _ in range(args.warmup):
            m(derm, clinic)
        prof = profiler.Profiler(only={"conv_gemm_128x128"})
        torch.cuda.synchronize()
        ops.set_profiler(prof)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            preds = m(derm, clinic)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        ops.set_profiler(None)
    dom = prof.summary().get("conv_gemm_128x128", {"flops": 0.0, "ms": 0.0, "launches": 0})
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    print(json.dumps({
        "metric": "SM3 inference pairs/sec (224x224, ResNet-50 x2 + multi-label transformer heads)",
        "value": round(B * args.steps / elapsed, 2), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"inference.py Model.forward (eval), batch {B}, {S}x{S}", "global_batch": B,
                   "parallelism": "dp1", "checksum": round(float(sum(p.float().abs().sum() for p in preds)), 4)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None,
                     "kernel": "conv_igemm_kernel<bf16_t,128,128,2,2,*> with the conv+evalBN+ReLU epilogue"}}), flush=True)


def mlc_train_bench(args):
    """Secondary line (BASELINE.json configs[3]): the multi-label DeepCluster step of tools/mlc_train.py:236-262 --
    frozen HIP encoders in eval mode (fused conv+BN epilogue), label projectors / 8-token transformer layer / prototype
    heads training on csrc/heads_train.hip, pseudo-label cross-entropy / 8, AdamW; plus the per-epoch spherical k-means
    (cluster_memory) timed apart over a 2048-sample memory bank."""
    import importlib.util
    from sm3hip import ops, profiler
    spec = importlib.util.spec_from_file_location("sm3_mlc_train", os.path.join(ROOT, "skin-sm3_amd", "tools", "mlc_train.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    from src.models.projector import MultiLabelProjector4
    from src.models.simclr import SimCLRSkinV32
    dev = torch.device("cuda", 0)
    torch.manual_seed(3407)
    B, S = (128 if args.batch == 256 else args.batch), args.img
    margs = mt.get_parser().parse_args(["--data-name", "synthetic", "--data-path", "-", "-b", str(B), "--mlc-proj-dim", "512",
                                       "--sa-dim-ff", "128"])  # run.sh:39-47
    ex = SimCLRSkinV32(arch="resnet50", proj_dim=128)
    ex.derm_backbone.projector = ex.clinic_backbone.projector = ex.cross_proj = None
    ex.sm3_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    for p in ex.parameters():
        p.requires_grad = False
    m = mt.Model(ex, MultiLabelProjector4(4096, 512, 8), 512, False, 1, 128, 0.1).to(dev)
    m.eval()
    m.projectors.train(); m.mlc_sa.train(); m.prototypes.train()
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=5e-2)
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100)
    g = torch.Generator(device=dev).manual_seed(3407)
    derm = torch.randn(B, 3, S, S, device=dev, generator=g)
    clinic = torch.randn(B, 3, S, S, device=dev, generator=g)
    assign = [torch.randint(0, pr.weight.size(0), (B,), device=dev, generator=g) for pr in m.prototypes]

    def step():
        _, preds = m(derm, clinic)
        loss = sum(crit(p / margs.temperature, a) for p, a in zip(preds, assign)) / 8
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    prof = profiler.Profiler(only={"conv_gemm_128x128"})
    torch.cuda.synchronize()
    ops.set_profiler(prof)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)
    # the per-epoch clustering: 8 spherical k-means (10 iterations each) over a 2048 x 512 memory bank
    nmem = 2048
    idx = torch.arange(nmem, device=dev)
    emb = torch.nn.functional.normalize(torch.randn(nmem, 512, device=dev, generator=g), dim=1)
    margs.world_size, margs.rank = 1, 0
    gk = torch.Generator().manual_seed(3407)
    torch.cuda.synchronize()
    tk = time.perf_counter()
    for pr in m.prototypes:
        mt.cluster_memory(margs, pr, pr.weight.size(0), idx, emb, generator=gk)
    torch.cuda.synchronize()
    kmeans_ms = 1e3 * (time.perf_counter() - tk)
    dom = prof.summary().get("conv_gemm_128x128", {"flops": 0.0, "ms": 0.0, "launches": 0})
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    print(json.dumps({
        "metric": "SM3 multi-label DeepCluster pairs/sec (224x224, frozen ResNet-50 x2 + transformer heads training)",
        "value": round(B * args.steps / elapsed, 2), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"tools/mlc_train.py step (run.sh:39-47: v4 projectors 512, sa_dim_ff 128, 1 head), "
                               f"batch {B}, {S}x{S}", "global_batch": B, "parallelism": "dp1", "loss": round(float(loss.detach()), 5),
                   "kmeans_ms_per_epoch": round(kmeans_ms, 2), "kmeans_bank": nmem},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None,
                     "kernel": "conv_igemm_kernel<bf16_t,128,128,2,2,*> with the conv+evalBN+ReLU epilogue"}}), flush=True)


def visible_gpus():
    """(count, source): compute nodes the kernel driver exposes (KFD topology: a node with SIMDs is a GPU), read from sysfs
    -- the parent of the rank processes never touches HIP -- capped by the shortest of HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one is set (a rank with LOCAL_RANK beyond that list would die in
    set_device).  (None, reason) when nothing is readable."""
    import glob
    n, src = None, "nothing readable"
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        try:
            n = 0
            for f in nodes:
                for line in open(f):
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
            src = "KFD topology in sysfs"
        except (OSError, ValueError):
            n, src = None, "KFD topology unreadable"
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None:
            continue
        listed = len([x for x in val.split(",") if x.strip() != ""])
        if n is None or listed < n:
            n, src = listed, f"{var}={val!r}"
    return n, src


def spawn_ranks(n, argv=None, extra_env=None, grace=5.0):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this same file, one rank
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), relay their
    stderr line by line with a "[rank r]" prefix, and watch them: as soon as ONE rank exits non-zero the others are
    terminated (they would otherwise sit in their next collective until the process-group timeout) and that exit code is
    returned.  The parent never initialises HIP; rank 0's stdout carries the ONE JSON line."""
    import socket
    import subprocess
    import threading
    have, have_src = visible_gpus()
    if os.environ.get("SM3_FORCE_DEVICE") is None and os.environ.get("SM3_BENCH_DRYRUN") != "1" and have is not None \
            and have < n:
        raise SystemExit(f"--gpus {n} but only {have} GPU(s) visible (from {have_src})")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = sys.argv[1:] if argv is None else list(argv)
    procs, relays = [], []

    def relay(r, pipe):
        for line in iter(pipe.readline, b""):
            sys.stderr.buffer.write(b"[rank %d] " % r + line)
            sys.stderr.buffer.flush()
        pipe.close()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        pr = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                              stdout=None if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(pr)
        th = threading.Thread(target=relay, args=(r, pr.stderr), daemon=True)
        th.start()
        relays.append(th)
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        time.sleep(0.05)
        for r in list(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    print(f"[bench] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr, flush=True)
                    break
    if rc != 0:  # children of this process, never re-exec'd: terminate, then kill what ignores it
        for r in live:
            procs[r].terminate()
        t_end = time.monotonic() + grace
        for r in live:
            try:
                procs[r].wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    for th in relays:
        th.join(timeout=2.0)
    return rc


def dry_run_rank(args):
    """TEST MODE (SM3_BENCH_DRYRUN=1, tests/test_bench_launcher.py): the host side of one rank of this benchmark on CPU
    tensors with the C ABI replaced by tests/fakelib.py -- launcher, rendezvous, per-rank data, collectives (gloo), JSON
    line; no kernel runs, so the numbers mean nothing and the line says "dry_run": true.  Never used by the driver."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fakelib import installed
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("SM3_BENCH_FAIL_RANK") == str(rank):
        raise RuntimeError(f"rank {rank}: injected start-up failure")
    if world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    with installed():
        torch.manual_seed(3407)
        model = SimCLRSkinV32("resnet50", None, 128, 0.1)
        model.sm3_dtype = torch.bfloat16
        trainer = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
        g = torch.Generator().manual_seed(3407 + rank)
        x = [torch.randn(2, 3, 32, 32, generator=g) for _ in range(4)]
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps)):
            trainer.step(x[:2], x[2:])
        if world > 1:
            dist.barrier()
        mine = time.perf_counter() - t0
        el = torch.tensor([mine], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        witness = dp_witness(world, "gloo", torch.device("cpu"), mine, max(1, args.steps), trainer)
    if rank == 0:
        print(json.dumps({"metric": "SM3 pretrain images/sec (paired 224x224)", "dry_run": True, "value": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"parallelism": f"dp{world}", "witness": witness}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def dp_witness(world, backend, dev, elapsed_s, steps, trainer=None):
    """What a multi-rank record must be able to prove (rank 0 returns the dict, the other ranks take part in the
    collectives): how many ranks the backend REALLY connected (an all-reduce of ones over the default group -- not
    WORLD_SIZE echoed back), which library that was, how the SyncBN statistics travelled, and every rank's own ms/step."""
    w = {"backend": backend, "rccl_ranks": None, "nccl_version": None, "syncbn_exchange": "none (single rank)",
         "ms_per_step_per_rank": [round(1e3 * elapsed_s / max(steps, 1), 3)]}
    if not (dist.is_available() and dist.is_initialized()):
        return w
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one)
    w["rccl_ranks"] = int(one.item())
    if backend == "nccl":
        try:
            w["nccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # a build without the query: say so, never guess
            w["nccl_version"] = f"unavailable ({type(e).__name__})"
    mine = torch.tensor([1e3 * elapsed_s / max(steps, 1)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    per = [round(float(x), 3) for x in every]
    w["ms_per_step_per_rank"] = per
    w["ms_per_step_min"], w["ms_per_step_max"] = min(per), max(per)
    p2p = getattr(trainer, "_p2p", None) if trainer is not None else None
    if p2p is not None:
        w["syncbn_exchange"] = "p2p"
        w["p2p_mailbox_memory"] = p2p.memory_kind
    elif trainer is not None and getattr(trainer, "sync_bn", False):
        w["syncbn_exchange"] = "rccl" if backend == "nccl" else backend
    return w


def main():
    args = parse()
    if args.workload == "linear_probe":
        return linear_probe_bench(args)
    if args.workload == "inference":
        return inference_bench(args)
    if args.workload == "mlc_train":
        return mlc_train_bench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if os.environ.get("SM3_BENCH_DRYRUN") == "1":
        return dry_run_rank(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # test knobs (a 2-rank rehearsal of this file on a one-GPU box: both ranks on device 0 over gloo)
    backend = os.environ.get("SM3_DIST_BACKEND", "nccl")
    if "SM3_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SM3_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SM3_BENCH_FORCE_DP=1 with one rank: the whole data-parallel path (RCCL communicators per lane, SyncBN statistic
    # all-reduces between the kernels, bucketed gradient all-reduces) on a one-GPU box -- every collective is the identity,
    # what is measured is their launch cost on the critical path (rehearsal knob, not a bench configuration)
    force_dp = world == 1 and os.environ.get("SM3_BENCH_FORCE_DP") == "1"
    if force_dp:
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=120)  # a rank that never arrives fails the run in two minutes, not in thirty
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    from sm3hip import ops, profiler
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32

    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    torch.manual_seed(3407)  # identical random-init weights on every rank (Kaiming fan_out, resnet.py:227-232)
    model = SimCLRSkinV32("resnet50", None, 128, 0.1, metadata_dim=args.metadata_dim or None)
    model.sm3_dtype = tdt
    model.to(dev)
    trainer = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0,  # run.sh:6 lr
                         global_negatives=args.global_negatives, target_momentum=args.target_momentum,
                         data_parallel=True if force_dp else None)

    g = torch.Generator(device=dev).manual_seed(3407 + rank)
    B, S = args.batch, args.img
    derm = [torch.randn(B, 3, S, S, device=dev, generator=g) for _ in range(2)]
    clinic = [torch.randn(B, 3, S, S, device=dev, generator=g) for _ in range(2)]
    if args.metadata_dim:
        meta = torch.randn(B, args.metadata_dim, device=dev, generator=g)
        _plain_step = trainer.step
        trainer.step = lambda d, c: _plain_step(d, c, metadata=meta)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.single_lane:
        trainer._engine().two_streams = False
    for _ in range(args.warmup):
        trainer.step(derm, clinic)
    # HIP events around every launch of the dominant kernel INSIDE the timed region -- of its last step only: an event record
    # is a marker packet in the lane's queue (rocprofv3 lists each as a 2.8 us __amd_rocclr_copyBuffer dispatch), 392 of them
    # per instrumented step, and with every step instrumented they cost the timed region itself ~1 % (SM3_BENCH_REGION_EVENTS=
    # all | last | none for the A/B; the serialised extra step below is what `achieved` comes from either way)
    region_events = os.environ.get("SM3_BENCH_REGION_EVENTS", "last")
    prof = profiler.Profiler(only={"conv_gemm_128x128"})
    sync()
    if region_events == "all":
        ops.set_profiler(prof)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if region_events == "last" and i == args.steps - 1:
            ops.set_profiler(prof)
        loss = trainer.step(derm, clinic)
    sync()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)
    loss_val = float(loss)
    witness = dp_witness(world, backend if (world > 1 or force_dp) else "none", dev, elapsed, args.steps, trainer)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    pairs_per_s = world * B * args.steps / elapsed

    # SM3_BENCH_AB_P2P=1 (multi-rank runs, opt-in): the same timed loop once more with the SyncBN statistics going through the
    # hipIpc mailboxes (csrc/p2p.hip) instead of RCCL; rank 0 then prints a SECOND, line-compatible record after the
    # contract's one -- the first multi-GPU run can A/B the two exchanges in one launch.  Never on by default.
    ab_record = None
    if os.environ.get("SM3_BENCH_AB_P2P") == "1" and (world > 1 or force_dp) and trainer.sync_bn \
            and getattr(trainer, "_p2p", None) is None:
        os.environ["SM3_SYNCBN_P2P"] = "1"
        trainer._engine().__dict__["_explicit_sync"] = None
        trainer._engine()  # re-wires the statistics exchange (maps the mailboxes: a collective over every rank)
        for _ in range(args.warmup):
            trainer.step(derm, clinic)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            loss2 = trainer.step(derm, clinic)
        sync()
        el2 = time.perf_counter() - t1
        trainer.check()
        w2 = dp_witness(world, backend, dev, el2, args.steps, trainer)
        t2 = torch.tensor([el2], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        el2 = float(t2)
        ab_record = {"metric": f"SM3 pretrain images/sec (paired {S}x{S})", "value": round(world * B * args.steps / el2, 2),
                     "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                     "ms_per_step": round(1e3 * el2 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                     "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                     "config": {"workload": "as the first line; SyncBN statistics exchanged peer-to-peer (SM3_SYNCBN_P2P=1)",
                                "global_batch": B * world, "parallelism": f"dp{world}", "loss": round(float(loss2), 5),
                                "witness": w2}}
        trainer.close()
        os.environ["SM3_SYNCBN_P2P"] = "0"
        trainer._engine()

    # The timed steps run the derm and clinic branches on two HIP streams, so a launch of the dominant kernel
    # shares the chip with the other lane's kernels and its event-bracketed duration there is NOT the kernel's own
    # speed.  `achieved` therefore comes from one extra, untimed step with the lanes serialised (same process, same
    # HIP-event bracketing on the launch stream), which is also what rocprofv3 --kernel-trace sees (it serialises
    # dispatches); the in-region figure is reported beside it.
    in_region = prof.summary().get("conv_gemm_128x128", {"flops": 0.0, "ms": 0.0, "launches": 0})
    eng = trainer._engine()
    eng.two_streams = False
    trainer.step(derm, clinic)
    iso = profiler.Profiler(only={"conv_gemm_128x128"})
    torch.cuda.synchronize()
    ops.set_profiler(iso)
    trainer.step(derm, clinic)
    torch.cuda.synchronize()
    ops.set_profiler(None)
    eng.two_streams = not args.single_lane
    dom = iso.summary().get("conv_gemm_128x128", {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0})
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    traffic, traffic_source = replayed_traffic(args.dtype, B, S)
    # The kernel's 432 launches per step straddle the ridge (peak FLOP/s / 8 TB/s = 312 FLOP/B at bf16): split them
    # by algorithmic intensity and price each side against its own roof (extra keys; `achieved`/`frac` above them
    # stay the all-launch MFMA figures the contract asks for).
    ridge = peak * 1e12 / HBM_PEAK_BYTES
    reg = {"mfma": [0.0, 0.0, 0.0, 0], "hbm": [0.0, 0.0, 0.0, 0]}
    for tag, fl, nb, ev0, ev1 in iso.records:
        r = reg["mfma" if fl / max(nb, 1.0) >= ridge else "hbm"]
        r[0] += fl; r[1] += nb; r[2] += ev0.elapsed_time(ev1) * 1e-3; r[3] += 1
    by_regime = {
        "mfma_bound_launches": {"launches": reg["mfma"][3], "achieved_TFLOPs": round(reg["mfma"][0] / max(reg["mfma"][2], 1e-9) / 1e12, 1),
                                "frac_of_mfma_peak": round(reg["mfma"][0] / max(reg["mfma"][2], 1e-9) / (peak * 1e12), 4)},
        "hbm_bound_launches": {"launches": reg["hbm"][3], "achieved_GBs": round(reg["hbm"][1] / max(reg["hbm"][2], 1e-9) / 1e9, 0),
                               "frac_of_hbm_peak": round(reg["hbm"][1] / max(reg["hbm"][2], 1e-9) / HBM_PEAK_BYTES, 4)},
        "ridge_flop_per_byte": round(ridge, 1)}
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                "frac_in_timed_region_two_lanes": round(in_region["flops"] / max(in_region["ms"] * 1e-3, 1e-9) / 1e12 / peak, 4),
                "by_regime": by_regime,
                "kernel": f"conv_igemm_kernel<{ {'bf16': 'bf16_t', 'f16': 'f16_t', 'f32': 'float'}[args.dtype] },128,128,*>",
                "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)),
                "launches_per_step": dom["launches"],
                "avg_launch_us": round(1e3 * dom["ms"] / max(dom["launches"], 1), 2),
                "avg_launch_us_in_timed_region_two_lanes": round(1e3 * in_region["ms"] / max(in_region["launches"], 1), 2),
                "avg_launch_gflop": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 3),
                "whole_step_mfma_frac": round(pairs_per_s / world * (FLOP_PER_PAIR_224 * (S / 224.0) ** 2) / (peak * 1e12), 4)}

    peak_gb = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)  # of the headline mode (read before the other-mode loops)
    if args.breakdown and rank == 0:
        full = profiler.Profiler()
        ops.set_profiler(full)
        trainer.step(derm, clinic)
        torch.cuda.synchronize()
        ops.set_profiler(None)
        with open(args.breakdown, "w") as f:
            f.write(full.format_table(f"per-kernel-class breakdown of one step, B={B}, {S}x{S}, {args.dtype}"))

    # The same step in the other two arithmetic modes, next to the headline (bf16 = BASELINE.json configs[1]): fp16 +
    # dynamic loss scaling is the reference's own AMP recipe (run.sh:10, backbone_train.py:27,98,480); exact-f32 MFMA is the
    # mode that meets north_star's 1e-3 loss / logit tolerance.  Short loops (same inputs, fresh weights), N=1 only.
    other_modes = None
    plain = not (args.global_negatives or args.metadata_dim or args.target_momentum or args.single_lane)
    if world == 1 and args.dtype == "bf16" and plain and not args.no_other_dtypes and not args.no_cpu_baseline:
        del trainer, model, eng
        torch.cuda.empty_cache()
        other_modes = {}
        for name, odt, nw, ns in (("f16", torch.float16, 6, 15), ("f32", torch.float32, 1, 3)):
            torch.manual_seed(3407)
            om = SimCLRSkinV32("resnet50", None, 128, 0.1)
            om.sm3_dtype = odt
            om.to(dev)
            otr = SM3Trainer(om, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
            for _ in range(nw):
                otr.step(derm, clinic)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(ns):
                otr.step(derm, clinic)
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t1
            other_modes[name + "_pairs_per_s"] = round(B * ns / dt_, 1)
            other_modes[name + "_steps_timed"] = ns
            del otr, om
            torch.cuda.empty_cache()
        other_modes["note"] = ("same step, same inputs: f16 = fp16 storage + f16 MFMA + device-side dynamic loss scaling "
                               "(the reference's AMP type); f32 = exact-f32 MFMA parity mode (loss / logits within 1e-3 "
                               "of the reference)")

    if rank == 0:
        out = {
            "metric": f"SM3 pretrain images/sec (paired {S}x{S})",
            "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SM3 pretrain step, SimCLRSkinV32(resnet50 x2), synthetic {S}x{S} derm+clinical "
                                   f"pairs, batch {B}/GPU, style 0, AdamW, random-init weights",
                       "global_batch": B * world, "parallelism": f"dp{world}", "encoder_images_per_s": round(4 * pairs_per_s, 1),
                       "negatives": "global (all-gather)" if args.global_negatives else "local (reference)",
                       "extensions": {"metadata_dim": args.metadata_dim, "target_momentum": args.target_momentum},
                       "loss": round(loss_val, 5),
                       "peak_hbm_allocated_gb": peak_gb,
                       # runtime switches in force (kernel arguments in device memory: set at the top of this file) and the
                       # arithmetic forms of the step: fixed-order weight gradients, 16-bit image stem
                       "runtime": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                                   "deterministic_weight_gradients": os.environ.get("SM3_WGRAD_DET", "1") != "0",
                                   "stem_16bit_image": os.environ.get("SM3_STEM16", "1") != "0" and args.dtype != "f32"},
                       # what ran, measured rather than echoed: ranks counted by an all-reduce of ones, the collective
                       # library's version, how the SyncBN statistics travelled, every rank's own ms/step
                       "witness": witness},
            "roofline": roofline,
        }
        if other_modes:
            out["config"]["other_modes"] = other_modes
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, S, args.cpu_steps)
        print(json.dumps(out), flush=True)
        if ab_record is not None:
            print(json.dumps(ab_record), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
