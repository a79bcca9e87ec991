"""Does a gather-GEMM with fewer registers per wave share a CU better with an HBM-bound kernel of the other lane?  (DESIGN.md
section 7 / 10: the 128-column kernels hold 110 - 127 registers per wave x 4 waves per SIMD = the whole register file; the
64-column instantiations of the same kernels hold 74 - 83.)  The same 3x3 convolutions on 128- and on 64-column tiles
(SM3_CONV_FORCE_NARROW=1), alone and beside a stream of BatchNorm backward-apply passes; what counts is the time of the PAIR
against the sum of the two alone."""
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


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(pairs):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in (s1, s2): st.wait_stream(torch.cuda.current_stream())
    for fn, n, st in pairs:
        with torch.cuda.stream(st), ops.stream_scope():
            for _ in range(n): fn()
    for st in (s1, s2): torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


print(__doc__)
bn = bn_apply_job(1605632, 64)
timed([(bn, 4, s1)])
for name, H, C in (("3x3 256 -> 256, 14 x 14, 512 images", 14, 256), ("3x3 128 -> 128, 28 x 28, 512 images", 28, 128)):
    conv = conv_job(512, H, C, C, 3)
    print(f"\n{name}")
    for narrow in ("0", "1"):
        os.environ["SM3_CONV_FORCE_NARROW"] = narrow
        nc = 24
        timed([(conv, 3, s1)])
        ta = min(timed([(conv, nc, s1)]) for _ in range(3))
        # as many BatchNorm passes as take about the convolutions' time alone
        tb1 = min(timed([(bn, 8, s2)]) for _ in range(3)) / 8
        nb = max(1, int(round(ta / tb1)))
        tb = min(timed([(bn, nb, s2)]) for _ in range(3))
        tp = min(timed([(conv, nc, s1), (bn, nb, s2)]) for _ in range(3))
        print(f"  {'64' if narrow == '1' else '128'}-column tiles: {nc} convolutions alone {ta:7.3f} ms ({ta / nc * 1e3:6.1f} us each); {nb} BatchNorm passes alone "
              f"{tb:7.3f} ms; together {tp:7.3f} ms = {tp / (ta + tb):.3f} of the sum   [0.5 = perfect overlap, 1.0 = none]")
os.environ["SM3_CONV_FORCE_NARROW"] = "0"
