"""Does chunking a BN-apply -> consumer-conv pair by sub-batches (so that y stays in the 256 MiB Infinity Cache between its
write and its read) beat the whole-batch launches?  Rows are batch-major NHWC, so a sub-batch is a contiguous row range."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops
dev = torch.device("cuda:0"); dt, code = torch.bfloat16, 1
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (N, H, Ci, Co) in [(512, 56, 256, 64), (512, 56, 64, 256), (512, 28, 512, 128), (512, 28, 128, 512), (512, 14, 1024, 256)]:
    rows = N * H * H
    xo = torch.randn(rows, Ci, device=dev).to(dt)
    y = torch.empty_like(xo)
    scale, shift = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    w = (torch.randn(Co, Ci, device=dev) * 0.05).to(dt)
    z = torch.empty(rows, Co, dtype=dt, device=dev)
    line = f"y {rows*Ci*2/1e6:.0f} MB z {rows*Co*2/1e6:.0f} MB Ci{Ci}->Co{Co}:"
    for chunks in (1, 2, 4, 8, 16, 32):
        n = N // chunks
        r = n * H * H
        d = ops.fwd_desc(code, n, H, H, Ci, Co, 1, 1, 0)
        prow = ops.conv_partial_rows(d)
        part = torch.empty(chunks * prow * 2 * Co, device=dev)
        def pair():
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.bn_act(code, xo[sl], scale, shift, None, True, y[sl], r, Ci)
                ops.conv_gemm(d, y[sl], w, z[sl], None, part[c * prow * 2 * Co:(c + 1) * prow * 2 * Co])
        def separate():
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.bn_act(code, xo[sl], scale, shift, None, True, y[sl], r, Ci)
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.conv_gemm(d, y[sl], w, z[sl], None, part[c * prow * 2 * Co:(c + 1) * prow * 2 * Co])
        line += f"  x{chunks}: {timeit(pair):.0f}/{timeit(separate):.0f}"
    print(line + "  us (interleaved/separate)", flush=True)

print("backward-like chain: apply(a->g) ; dgrad(g->dx) ; wgrad(x_in, g -> dw)")
for (N, H, C0, C1) in [(512, 56, 64, 256), (512, 56, 256, 64), (512, 28, 128, 512), (512, 28, 512, 128)]:
    rows = N * H * H
    a = torch.randn(rows, C1, device=dev).to(dt)
    g = torch.empty_like(a)
    x_in = torch.randn(rows, C0, device=dev).to(dt)
    dx = torch.empty_like(x_in)
    scale, shift = torch.ones(C1, device=dev), torch.zeros(C1, device=dev)
    wd = (torch.randn(C0, C1, device=dev) * 0.05).to(dt)
    line = f"g {rows*C1*2/1e6:.0f} MB x {rows*C0*2/1e6:.0f} MB C0={C0} C1={C1}:"
    for chunks in (1, 2, 4, 8, 16, 32):
        n = N // chunks
        r = n * H * H
        dd = ops.fwd_desc(code, n, H, H, C1, C0, 1, 1, 0)      # dgrad of a 1x1 = 1x1 conv C1 -> C0
        dwd = ops.fwd_desc(code, n, H, H, C0, C1, 1, 1, 0)     # the forward conv whose weight gradient is taken
        dw = torch.zeros(C1 * dwd.w_row_stride, device=dev)
        def chain():
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.bn_act(code, a[sl], scale, shift, None, False, g[sl], r, C1)
                ops.conv_gemm(dd, g[sl], wd, dx[sl], None, None)
                ops.conv_wgrad(dwd, x_in[sl], g[sl], dw)
        def separate():
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.bn_act(code, a[sl], scale, shift, None, False, g[sl], r, C1)
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.conv_gemm(dd, g[sl], wd, dx[sl], None, None)
            for c in range(chunks):
                sl = slice(c * r, (c + 1) * r)
                ops.conv_wgrad(dwd, x_in[sl], g[sl], dw)
        line += f"  x{chunks}: {timeit(chain):.0f}/{timeit(separate):.0f}"
    print(line + "  us (interleaved/separate)", flush=True)
