"""s_memtime timeline of the SHIPPING weight-gradient ring loop: a second build of the library with
-DSM3_STAMP (scratch/build_stamp.sh -> scratch/_stamp/libsm3hip_stamp.so; the product library contains no stamp) records, per
wave and K-step of 32 pixels,
  wait    = counted s_waitcnt vmcnt: this wave's LDS-DMA pieces of the stage it is about to compute have landed
  barrier = s_barrier: everyone's have, and everyone is done with the stage that is overwritten next
  issue   = LDS-DMA issue of the stage NST - 1 steps ahead
  compute = transposing fragment reads (ds_read_b64_tr_b16) + MFMA issue of this stage
plus prologue (entry -> first stages in flight) and epilogue (loop end -> exit: K-group hand-over through LDS, slab stores or
float atomics).  Usage (GPU box, repo root):  SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so python3 scratch/stamp_wgrad.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops  # noqa: E402

lib = ctypes.CDLL(os.environ["SM3_LIBRARY"])
lib.sm3_debug_set_wgrad_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long]
dev = torch.device("cuda:0")
dt, code = torch.bfloat16, ops.dtype_code(torch.bfloat16)
CAP = 1 << 16


def run(name, N, H, W, Ci, Co, k, slabs):
    d = ops.fwd_desc(code, N, H, W, Ci, Co, k, 1, k // 2)
    M = N * H * W
    x = torch.randn(M, Ci, device=dev).to(dt)
    dy = torch.randn(M, Co, device=dev).to(dt)
    dw = torch.zeros(Co, k * k * Ci, device=dev)
    cap = max(1, min(ops.SLAB_CAP, (1 << 24) // (Co * Ci)))  # engine._slab_buf's rule
    sl = torch.empty(2 * cap * Co * Ci, device=dev) if slabs else None
    buf = torch.zeros(CAP * 16, dtype=torch.int64, device=dev)

    def launch():
        if slabs:
            return ops.conv_wgrad_slabs(d, x, dy, sl, views=2, cap=cap)
        ops.conv_wgrad(d, x, dy, dw)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    assert lib.sm3_debug_set_wgrad_stamps(buf.data_ptr(), CAP) == 0
    times = []
    for _ in range(3):
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ns = launch()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    lib.sm3_debug_set_wgrad_stamps(None, 0)
    r = buf.cpu().numpy().astype(np.uint64).reshape(CAP, 16)
    r = r[(r[:, 3] > 0) & (r[:, 8] > 0)]
    nst = r[:, 8].astype(np.float64)
    f = lambda a: (float(np.median(a)), float(np.percentile(a, 10)), float(np.percentile(a, 90)))
    seg = [r[:, 4 + i].astype(np.float64) / nst for i in range(4)]
    pro = (r[:, 1] - r[:, 0]).astype(np.float64)
    loop = (r[:, 2] - r[:, 1]).astype(np.float64)
    epi = (r[:, 3] - r[:, 2]).astype(np.float64)
    life = (r[:, 3] - r[:, 0]).astype(np.float64)
    clk = float(life.sum() / np.maximum((r[:, 11] - r[:, 10]).astype(np.float64).sum(), 1.0)) * 0.1
    ms = min(times)
    tf = 2.0 * M * Co * k * k * Ci / (ms * 1e-3) / 1e12
    gb = 2.0 * M * (Ci + Co) / (ms * 1e-3) / 1e9
    nwg = len(np.unique(r[:, 9]))
    print(f"\n{name}: M={M} Ci={Ci} Co={Co} k={k} {'slabs (' + str(ns) + ' per view)' if slabs else 'atomics'}; {nwg} workgroups, {len(r)} waves, "
          f"{int(np.median(nst))} K-steps per wave; launch {ms * 1e3:.0f} us = {tf:.0f} TFLOP/s, {gb:.0f} GB/s algorithmic (stamped build; "
          f"{[round(t * 1e3) for t in times]} us); in-kernel clock ~{clk:.2f} GHz")
    print("  cycles per K-step and wave, median (p10 .. p90):")
    tot = 0.0
    for nm, a in zip(("wait (counted vmcnt)", "barrier", "issue (LDS-DMA of the stage ahead)", "compute (tr-reads + MFMA issue)"), seg):
        m, lo, hi = f(a)
        tot += m
        print(f"    {nm:40s} {m:8.0f}  ({lo:.0f} .. {hi:.0f})")
    print(f"    {'sum':40s} {tot:8.0f}")
    for nm, a in (("prologue", pro), ("ring loop", loop), ("epilogue (hand-over + stores / atomics)", epi), ("wave lifetime", life)):
        m, lo, hi = f(a)
        print(f"  {nm:42s} {m:9.0f}  ({lo:.0f} .. {hi:.0f}) cycles")


print(__doc__.split("Usage")[0])
run("layer3 conv3 moments P = dz^T y2 (1x1 256 -> 1024)", 512, 14, 14, 256, 1024, 1, True)
run("layer3 conv1 (1x1 1024 -> 256)", 512, 14, 14, 1024, 256, 1, False)
run("layer2 conv3 moments (1x1 128 -> 512)", 512, 28, 28, 128, 512, 1, True)
run("layer2 conv1 (1x1 512 -> 128)", 512, 28, 28, 512, 128, 1, False)
run("layer1 conv3 moments (1x1 64 -> 256)", 512, 56, 56, 64, 256, 1, True)
run("layer4 conv3 moments (1x1 512 -> 2048)", 512, 7, 7, 512, 2048, 1, True)
run("layer3 conv2 (3x3 256 -> 256, tap-shifted one-stage kernel)", 512, 14, 14, 256, 256, 3, False)
