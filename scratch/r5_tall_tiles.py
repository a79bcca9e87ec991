"""256 x 64 tiles for the plain-epilogue 3x3 forward launches (SM3_CONV_HALO_TALL=1) against the shipping 128 x 128 / 128 x 64
halo tiles: same bits (output and BatchNorm partial rows), time alone, and time beside a second stream running the same
convolution / an HBM-bound BatchNorm pass."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; code = ops.dtype_code(dt)


def conv_job(N, H, Ci, Co, k):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, k // 2); M = N * H * H
    g = torch.Generator(device=dev).manual_seed(N + H + Ci)
    x = torch.randn(M, Ci, device=dev, generator=g).to(dt); w = (torch.randn(Co, k * k * Ci, device=dev, generator=g) * 0.05).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev); part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    return (lambda: ops.conv_gemm(d, x, w, y, None, part)), y, part


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
for name, N, H, C in (("3x3 256 -> 256, 14 x 14, 512 images", 512, 14, 256), ("3x3 128 -> 128, 28 x 28, 512 images", 512, 28, 128),
                      ("3x3 512 -> 512, 7 x 7, 512 images", 512, 7, 512), ("3x3 256 -> 256, 14 x 14, 101 images (M % 256 = 172)", 301, 14, 256)):
    conv, y, part = conv_job(N, H, C, C, 3)
    outs = {}
    print(f"\n{name}")
    for tall in ("0", "1"):
        os.environ["SM3_CONV_HALO_TALL"] = tall
        y.fill_(float("nan")); part.zero_()
        conv(); torch.cuda.synchronize()
        outs[tall] = (y.clone(), part.clone())
        nc = 24
        timed([(conv, 3, s1)])
        ta = min(timed([(conv, nc, s1)]) for _ in range(3))
        tcc = min(timed([(conv, nc, s1), (conv, nc, s2)]) for _ in range(3))
        tb1 = min(timed([(bn, 8, s2)]) for _ in range(3)) / 8
        nb = max(1, int(round(ta / tb1)))
        tb = min(timed([(bn, nb, s2)]) for _ in range(3))
        tp = min(timed([(conv, nc, s1), (bn, nb, s2)]) for _ in range(3))
        print(f"  {'256 x 64 ' if tall == '1' else 'shipping '} tiles: alone {ta / nc * 1e3:6.1f} us; two streams of the same convolution "
              f"{tcc / (2 * nc) * 1e3:6.1f} us each; beside {nb} BatchNorm passes ({tb:6.3f} ms alone) {tp:6.3f} ms = {tp / (ta + tb):.3f} of the sum")
    print("  bit-identical output:", bool(torch.equal(outs["0"][0], outs["1"][0])), " partial rows:", bool(torch.equal(outs["0"][1], outs["1"][1])))
os.environ["SM3_CONV_HALO_TALL"] = "0"
