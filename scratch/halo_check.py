"""kVarHalo (halo-resident A image of the stride-1 3x3 launches) against the general gather on the same inputs: forward (+BN
partial sums) and data gradient with the fused BN-backward phase 1; then per-shape timing of both."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops
D = torch.device("cuda:0"); dt = torch.bfloat16; code = ops.dtype_code(dt)

def run(N, H, W, Ci, Co, seed):
    gg = torch.Generator().manual_seed(seed)
    d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1); M = N * H * W
    x = torch.randn(M, Ci, generator=gg).to(dt).to(D)
    w = (torch.randn(Co, 9 * Ci, generator=gg) / math.sqrt(9 * Ci)).to(dt).to(D)
    y = torch.empty(M, Co, dtype=dt, device=D); part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=D)
    ops.conv_gemm(d, x, w, y, None, part)
    outs = [y, part]
    descs, full = ops.dgrad_descs(code, N, H, W, Ci, Co, 3, 1, 1)
    dy = torch.randn(M, Co, generator=gg).to(dt).to(D)
    wdg = (torch.randn(Ci, 9 * Co, generator=gg) / math.sqrt(9 * Co)).to(dt).to(D)
    add = torch.randn(M, Ci, generator=gg).to(dt).to(D); bnx = torch.randn(M, Ci, generator=gg).to(dt).to(D)
    msk = torch.randint(0, 256, (M * Ci // 8,), generator=gg, dtype=torch.uint8).to(D)
    mean, istd = torch.randn(Ci, generator=gg).to(D), (torch.rand(Ci, generator=gg) + 0.5).to(D)
    for with_x in (True, False):
        dz = torch.empty(M, Ci, dtype=dt, device=D); total = sum(ops.conv_partial_rows(dd) for dd in descs)
        fp = torch.zeros(total * 2 * Ci, device=D); off = 0
        for dd in descs: off += ops.conv_dgrad_bnfuse(dd, dy, wdg, dz, add, msk, bnx if with_x else None, mean, istd, fp, off)
        outs += [dz, fp]
    torch.cuda.synchronize()
    return outs

cases = [(3, 14, 14, 256, 256), (2, 28, 28, 128, 128), (2, 56, 56, 64, 64), (5, 7, 7, 512, 512), (3, 10, 14, 128, 256), (1, 3, 5, 64, 64)]
bad = 0
os.environ["SM3_CONV_DEEP"] = "0"  # (small grids would otherwise take the 4-stage kernel, whatever SM3_CONV_HALO says)
for ci, c in enumerate(cases):
    os.environ["SM3_CONV_HALO"] = "0"; a = run(*c, seed=5 + ci)
    os.environ["SM3_CONV_HALO"] = os.environ.get("HALO_MODE", "1"); b = run(*c, seed=5 + ci)
    for i, (u, v) in enumerate(zip(a, b)):
        u, v = u.double(), v.double(); scale = float(u.abs().max()) + 1e-30
        err = float((u - v).abs().max()) / scale; frac = int((u != v).sum())
        ok = err < (2 ** -7 if i % 2 == 0 else 2e-3)
        bad += not ok
        print(c, ["y", "part", "dz_x", "fp_x", "dz", "fp"][i], f"rel max diff {err:.2e}  differing {frac} of {u.numel()}", "" if ok else "  <-- BAD", flush=True)
os.environ.pop("SM3_CONV_DEEP")
print("BAD" if bad else "ALL OK")
if bad: sys.exit(1)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (N, H, Ci) in [(512, 14, 256), (512, 28, 128), (512, 56, 64), (512, 7, 512)]:
    d = ops.fwd_desc(code, N, H, H, Ci, Ci, 3, 1, 1); M = N * H * H
    x = torch.randn(M, Ci, device=D).to(dt); w = (torch.randn(Ci, 9 * Ci, device=D) * 0.02).to(dt)
    y = torch.empty(M, Ci, dtype=dt, device=D); part = torch.zeros(ops.conv_partial_rows(d) * 2 * Ci, device=D)
    descs, full = ops.dgrad_descs(code, N, H, H, Ci, Ci, 3, 1, 1)
    add = torch.randn(M, Ci, device=D).to(dt); msk = torch.randint(0, 256, (M * Ci // 8,), dtype=torch.uint8, device=D)
    mean, istd = torch.randn(Ci, device=D), torch.rand(Ci, device=D) + 0.5
    fp = torch.zeros(ops.conv_partial_rows(descs[0]) * 2 * Ci, device=D); dz = torch.empty_like(y)
    fl = 2.0 * M * 9 * Ci * Ci
    row = f"3x3 {Ci:4d}ch H={H:2d} M={M:8d}:"
    for mode in ("0", "1"):
        os.environ["SM3_CONV_HALO"] = mode
        t1 = timeit(lambda: ops.conv_gemm(d, x, w, y, None, part))
        t2 = timeit(lambda: ops.conv_dgrad_bnfuse(descs[0], y, w, dz, add, msk, x, mean, istd, fp, 0))
        row += f"   halo={mode}: fwd {t1*1e3:7.1f} us {fl/t1/1e9:7.1f} TF | dgrad+fz {t2*1e3:7.1f} us {fl/t2/1e9:7.1f} TF"
    print(row, flush=True)
