"""Co-run efficiency of kernel pairs on two streams: time(X || Y) against time(X) + time(Y) alone, for MFMA-bound and HBM-bound
members of the step (is an anti-phase schedule of the two lanes worth building?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; code = ops.dtype_code(dt)

def conv_job(N, H, Ci, Co, k):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, k // 2); M = N * H * H
    x = torch.randn(M, Ci, device=dev).to(dt); w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev); part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    return lambda: ops.conv_gemm(d, x, w, y, None, part)

def bn_apply_job(rows, C):
    dz = torch.randn(rows, C, device=dev).to(dt); x = torch.randn(rows, C, device=dev).to(dt); dx = torch.empty_like(dz)
    mean, istd, gamma = torch.randn(2 * C, device=dev), torch.rand(2 * C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    gs = torch.randn(2 * 2 * C, dtype=torch.float64, device=dev); ls = gs.clone()
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    return lambda: ops.bn_bwd_apply(code, dz, x, mean, istd, gamma, gs, float(rows), ls, dg, db, dx, rows // 2, C, views=2)

def wgrad_job(N, H, Ci, Co, k):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, k // 2); M = N * H * H
    x = torch.randn(M, Ci, device=dev).to(dt); dy = torch.randn(M, Co, device=dev).to(dt)
    dw = torch.zeros(Co, k * k * Ci, device=dev)
    return lambda: ops.conv_wgrad(d, x, dy, dw)

jobs = {
    "conv3x3_256 (MFMA)": (conv_job(512, 14, 256, 256, 3), 12),
    "conv3x3_128 (MFMA)": (conv_job(512, 28, 128, 128, 3), 12),
    "wgrad3x3_256 (MFMA)": (wgrad_job(512, 14, 256, 256, 3), 10),
    "conv1x1_64->256 (HBM)": (conv_job(512, 56, 64, 256, 1), 6),
    "bn_bwd_apply_C64 (HBM)": (bn_apply_job(1605632, 64), 16),
    "bn_bwd_apply_C256 (HBM, small)": (bn_apply_job(100352, 256), 40),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def run(fn, n, st):
    with torch.cuda.stream(st), ops.stream_scope():
        for _ in range(n): fn()

def timed(pairs):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in (s1, s2): st.wait_stream(torch.cuda.current_stream())
    for fn, n, st in pairs: run(fn, n, st)
    for st in (s1, s2): torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)

alone = {}
for k, (fn, n) in jobs.items():
    timed([(fn, n, s1)]); alone[k] = min(timed([(fn, n, s1)]) for _ in range(3))
    print(f"alone  {k:34s} {n:3d} launches {alone[k]:7.3f} ms", flush=True)
names = list(jobs)
print("\npair: time together / (time A alone + time B alone)   [1.0 = no gain from co-running, 0.5 = perfect overlap]")
for i in range(len(names)):
    for j in range(i, len(names)):
        a, b = names[i], names[j]
        fa, na = jobs[a]; fb, nb = jobs[b]
        t = min(timed([(fa, na, s1), (fb, nb, s2)]) for _ in range(3))
        print(f"  {a:32s} || {b:32s} {t:7.3f} ms  ratio {t / (alone[a] + alone[b]):.3f}", flush=True)
