"""s_memtime timeline of the SHIPPING gather-GEMM K loop: a second build of the library with -DSM3_STAMP
(scratch/_stamp/libsm3hip_stamp.so: `hipcc -DSM3_STAMP` over the same sources; the product library contains no stamp) records,
per wave, the cycles between the segment boundaries of the 2-stage loop --
  issue   = tap bookkeeping + LDS-DMA issue of the next stage
  compute = 4 x (ds_read_b128 fragments + 16 MFMA issue)           (MFMAs retire asynchronously: issue-side time)
  drain   = s_waitcnt vmcnt(0): this wave's part of the next stage has landed
  barrier = s_barrier: everyone's has, and everyone is done reading this stage
plus prologue (entry -> first stage published) and epilogue (loop end -> exit).  Usage (GPU box, repo root):
  SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so python3 scratch/stamp_conv.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops  # noqa: E402

lib = ctypes.CDLL(os.environ["SM3_LIBRARY"])
lib.sm3_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long]
dev = torch.device("cuda:0")
dt, code = torch.bfloat16, ops.dtype_code(torch.bfloat16)


def run(name, N, H, W, Ci, Co, k, stats=True):
    d = ops.fwd_desc(code, N, H, W, Ci, Co, k, 1, k // 2)
    M = N * d.Ho * d.Wo
    x = torch.randn(N * H * W, Ci, device=dev).to(dt)
    w = (torch.randn(Co, k * k * Ci, device=dev) / (k * k * Ci) ** 0.5).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev)
    part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev) if stats else None
    waves = ((M + 127) // 128) * ((Co + 127) // 128) * 4
    buf = torch.zeros(waves * 24, dtype=torch.int64, device=dev)
    assert lib.sm3_debug_set_stamps(buf.data_ptr(), waves) == 0
    for _ in range(3):
        ops.conv_gemm(d, x, w, y, None, part)
    torch.cuda.synchronize()
    buf.zero_()
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv_gemm(d, x, w, y, None, part)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    lib.sm3_debug_set_stamps(None, 0)
    r = buf.cpu().numpy().astype(np.uint64).reshape(waves, 24)
    r = r[r[:, 3] > 0]
    ns = r[:, 8].astype(np.float64)
    ok = ns > 0
    r, ns = r[ok], ns[ok]
    f = lambda a: (float(np.median(a)), float(np.percentile(a, 10)), float(np.percentile(a, 90)))
    seg = [r[:, 4 + i].astype(np.float64) / ns for i in range(4)]
    loop = (r[:, 2] - r[:, 1]).astype(np.float64)
    pro = (r[:, 1] - r[:, 0]).astype(np.float64)
    epi = (r[:, 3] - r[:, 2]).astype(np.float64)
    ms = min(times)
    # in-kernel clock: s_memtime ticks per s_memrealtime tick (100 MHz) over every wave's lifetime
    clk = float((r[:, 3] - r[:, 0]).astype(np.float64).sum() / np.maximum((r[:, 11] - r[:, 10]).astype(np.float64).sum(), 1.0)) * 0.1
    tf = 2.0 * M * Co * k * k * Ci / (ms * 1e-3) / 1e12
    print(f"\n{name}: M={M} K={k*k}x{Ci} N={Co}; {len(r)} waves, {int(np.median(ns))} K-steps; launch {ms*1e3:.0f} us = {tf:.0f} TFLOP/s "
          f"(stamped build; three launches {[round(t * 1e3) for t in times]} us); in-kernel clock ~{clk:.2f} GHz")
    print("  cycles per K-step, median (p10 .. p90):")
    tot = 0.0
    for nm, a in zip(("issue (DMA of next stage)", "compute (4 x (4 ds_read_b128 + 4 MFMA))", "drain (s_waitcnt vmcnt(0))", "barrier"), seg):
        m, lo, hi = f(a)
        tot += m
        print(f"    {nm:40s} {m:8.0f}  ({lo:.0f} .. {hi:.0f})")
    print(f"    {'sum':40s} {tot:8.0f}   [4 x 4 = 16 MFMAs of 32 cycles = 512 cycles of MFMA pipe per K-step per wave; 2 waves per SIMD]")
    for nm, a in (("prologue (entry -> stage 0 published)", pro), ("K loop", loop), ("epilogue (loop end -> exit)", epi)):
        m, lo, hi = f(a)
        print(f"  {nm:42s} {m:9.0f}  ({lo:.0f} .. {hi:.0f}) cycles per wave")


print(__doc__.split("Usage")[0])
print("(SM3_CONV_SINGLE_STAGE_MAX = %s: the ONE-stage loop runs up to that many K-steps -- there 'issue' is the DMA issue of the next\n"
      " stage AFTER the barrier that ends the reads of this one, 'drain' the full landing latency, 'barrier' both barriers of a K-step;\n"
      " four workgroups per CU overlap each other instead of two stages inside one)" % os.environ.get("SM3_CONV_SINGLE_STAGE_MAX", "default (always)"))
run("layer3 conv2, 3x3 256->256 (lean forward + BN sums)", 512, 14, 14, 256, 256, 3)
run("layer2 conv2, 3x3 128->128", 512, 28, 28, 128, 128, 3)
run("layer4 conv2, 3x3 512->512", 512, 7, 7, 512, 512, 3)
run("layer3 conv1, 1x1 1024->256", 512, 14, 14, 1024, 256, 1)
