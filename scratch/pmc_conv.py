"""A few dispatches of the conv kernels for PMC collection (rocprofv3 --pmc ...)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops
N = 256
dev = torch.device("cuda:0"); dt, code = torch.bfloat16, 1
SHAPES = [(256,256,3,1,14),(64,256,1,1,56),(1024,256,1,1,14),(128,128,3,1,28)]
for (Ci, Co, k, s, H) in SHAPES:
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, k // 2)
    x = torch.randn(N, H, H, Ci, device=dev).to(dt)
    w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
    y = torch.empty(N, d.Ho, d.Wo, Co, dtype=dt, device=dev)
    dy = torch.randn(N, d.Ho, d.Wo, Co, device=dev).to(dt)
    dw = torch.zeros(Co, k * k * Ci, device=dev)
    part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    for _ in range(3):
        ops.conv_gemm(d, x, w, y, None, part)
        ops.conv_wgrad(d, x, dy, dw)
    torch.cuda.synchronize()
