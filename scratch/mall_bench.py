"""Does walking a consumer in the opposite direction of its producer turn HBM reads into Infinity-Cache hits?"""
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
for (N, H, Ci, Co) in [(256, 56, 256, 64), (256, 56, 64, 256), (256, 28, 512, 128), (256, 14, 1024, 256)]:
    rows = N * H * H
    xo = torch.randn(rows, Ci, device=dev).to(dt)
    y = torch.empty_like(xo)
    scale, shift = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    d = ops.fwd_desc(code, N, H, H, Ci, Co, 1, 1, 0)
    w = (torch.randn(Co, Ci, device=dev) * 0.05).to(dt)
    z = torch.empty(rows, Co, dtype=dt, device=dev)
    part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    res = {}
    for rev in ("0", "1"):
        os.environ["SM3_CONV_REVERSE"] = rev
        def pair():
            ops.bn_act(code, xo, scale, shift, None, True, y, rows, Ci)
            ops.conv_gemm(d, y, w, z, None, part)
        t_pair = timeit(pair)
        t_act = timeit(lambda: ops.bn_act(code, xo, scale, shift, None, True, y, rows, Ci))
        res[rev] = (t_pair, t_act)
    print(f"y {rows*Ci*2/1e6:.0f} MB  Ci{Ci}->Co{Co}: pair fwd-order {res['0'][0]:.1f}us, reversed consumer {res['1'][0]:.1f}us (bn_act alone {res['0'][1]:.1f}us)")
