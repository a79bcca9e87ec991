"""s_memtime phase timeline of the gather-GEMM launches (where does a short-K 1x1 launch spend its
workgroup's lifetime?).  Needs the -DSM3_STAMP build (see stamp_conv.py): marks per wave at
  setup   entry -> loader / fragment state ready            stage0  -> first stage published (DMA issue + landing + barrier)
  kloop   -> K loop done                                     request -> epilogue operand requests issued (EPI >= 2)
  staging -> accumulators converted + staged in LDS, barrier readback-> read-back loop done (waits for the requested
  fzred   -> fused BN-backward reduction done                           operands, per-row work, stores issued)
Usage (GPU box, repo root):  SM3_LIBRARY=scratch/_stamp/libsm3hip_stamp.so python3 scratch/stamp_phases.py"""
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
W = 24


def stamped(name, fn, waves, flops, nbytes, wg_waves=4):
    buf = torch.zeros(waves * W, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert lib.sm3_debug_set_stamps(buf.data_ptr(), waves) == 0
    times = []
    for _ in range(3):
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    lib.sm3_debug_set_stamps(None, 0)
    r = buf.cpu().numpy().astype(np.uint64).reshape(waves, W)
    r = r[r[:, 3] > 0]
    ent, ext = r[:, 0].astype(np.float64), r[:, 3].astype(np.float64)
    m = r[:, 12:20].astype(np.float64)
    f = lambda a: (float(np.median(a)), float(np.percentile(a, 10)), float(np.percentile(a, 90)))
    clk = float((ext - ent).sum() / np.maximum((r[:, 11] - r[:, 10]).astype(np.float64).sum(), 1.0)) * 0.1
    ms = min(times)
    print(f"\n{name}: launch {ms*1e3:.0f} us = {flops/ms/1e9:.0f} TFLOP/s, {nbytes/ms/1e6:.0f} GB/s algorithmic (stamped build); "
          f"{len(r)} waves, {int(np.median(r[:, 8]))} K-steps, in-kernel clock ~{clk:.2f} GHz")
    order = [("setup    (entry -> state ready)", ent, m[:, 0]), ("stage0   (first stage issue + landing + barrier)", m[:, 0], m[:, 1]),
             ("kloop    (K loop incl. later stages)", m[:, 1], m[:, 2]), ("request  (epilogue operand requests issued)", m[:, 2], m[:, 6]),
             ("staging  (cvt + stats + LDS staging + barrier)", m[:, 6], m[:, 3]), ("readback (operand wait, per-row work, stores issued)", m[:, 3], m[:, 4]),
             ("fzred    (fused BN-backward reduction)", m[:, 4], m[:, 5]), ]
    last = ent
    tot = f(ext - ent)[0]
    for nm, a, b in order:
        ok = (b > 0) & (a > 0)
        if ok.sum() < len(r) // 2:
            continue
        med, lo, hi = f((b - a)[ok])
        print(f"    {nm:56s} {med:8.0f}  ({lo:.0f} .. {hi:.0f})  {100 * med / tot:5.1f} %")
    print(f"    {'lifetime of a wave (entry -> exit)':56s} {tot:8.0f}  = {tot / clk / 1e3:.2f} us at the in-kernel clock")


def fwd_plain(N, H, Ci, Co):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, 1, 1, 0); M = N * H * H
    x = torch.randn(M, Ci, device=dev).to(dt); w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev); part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    waves = ((M + 127) // 128) * ((Co + 127) // 128) * 8
    stamped(f"forward 1x1 {Ci}->{Co} M={M} (EPI 1, BN sums)", lambda: ops.conv_gemm(d, x, w, y, None, part), waves,
            2.0 * M * Ci * Co, 2.0 * (M * Ci + M * Co))


def fwd_bnact(N, H, Ci, Co):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, 1, 1, 0); M = N * H * H
    x = torch.randn(M, Ci, device=dev).to(dt); w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev); res = torch.randn(M, Co, device=dev).to(dt)
    sc = torch.rand(2 * Co, device=dev) + 0.5; sh = torch.randn(2 * Co, device=dev)
    mask = torch.empty(M * Co // 8, dtype=torch.uint8, device=dev)
    waves = ((M + 127) // 128) * ((Co + 127) // 128) * 8
    stamped(f"forward 1x1 {Ci}->{Co} M={M} + BN affine + identity + ReLU + bits (EPI 2, two views)",
            lambda: ops.conv_bn_act_fused(d, x, w, sc, sh, res, True, y, mask, views=2), waves, 2.0 * M * Ci * Co,
            2.0 * (M * Ci + 2 * M * Co))


def dgrad_fz(N, H, Ci, Co):
    """data gradient of a 1x1 conv Co -> Ci channels forward: dy [M, Ci] x W -> [M, Co] + identity gradient, masked,
    sum(dz) per tile (the linear BatchNorm forms need no x)."""
    d = ops.fwd_desc(code, N, H, H, Ci, Co, 1, 1, 0); M = N * H * H
    dy = torch.randn(M, Ci, device=dev).to(dt); w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).to(dt)
    dz = torch.empty(M, Co, dtype=dt, device=dev); add = torch.randn(M, Co, device=dev).to(dt)
    mask = torch.randint(0, 256, (M * Co // 8,), dtype=torch.uint8, device=dev)
    part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    waves = ((M + 127) // 128) * ((Co + 127) // 128) * 8
    stamped(f"data gradient 1x1 K={Ci} -> N={Co} M={M} + identity gradient + ReLU mask + sum(dz) (EPI 3, two views)",
            lambda: ops.conv_dgrad_bnfuse(d, dy, w, dz, add, mask, None, None, None, part, 0, views=2,
                                          row_offset_view1=ops.conv_partial_rows(d) // 2),
            waves, 2.0 * M * Ci * Co, 2.0 * (M * Ci + 2 * M * Co))


print(__doc__.split("Usage")[0])
fwd_plain(512, 28, 512, 128)      # layer2 conv1
fwd_plain(512, 14, 1024, 256)     # layer3 conv1 (2-stage)
fwd_bnact(512, 28, 128, 512)      # layer2 conv3 fused
fwd_bnact(512, 14, 256, 1024)     # layer3 conv3 fused
dgrad_fz(512, 56, 64, 256)        # layer1 conv1 dgrad
dgrad_fz(512, 28, 128, 512)       # layer2 conv1 dgrad
dgrad_fz(512, 14, 256, 1024)      # layer3 conv1 dgrad
dgrad_fz(512, 7, 512, 2048)       # layer4 conv1 dgrad
