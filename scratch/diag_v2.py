import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops
N = 256
dev = torch.device("cuda:0"); dt, code = torch.bfloat16, 1
for shape in [(256,256,3,1,14), (512,512,3,1,7), (256,1024,1,1,14), (1024,256,1,1,14), (64,256,1,1,56)]:
    Ci, Co, k, s, H = shape
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, k // 2)
    x = torch.randn(N, H, H, Ci, device=dev).to(dt)
    w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
    y = torch.empty(N, d.Ho, d.Wo, Co, dtype=dt, device=dev)
    part = torch.empty((N * d.Ho * d.Wo + 127) // 128 * 2 * Co, device=dev)
    os.environ["SM3_CONV_V2"] = "1"
    for _ in range(3): ops.conv_gemm(d, x, w, y, None, part)
    torch.cuda.synchronize()
    os.environ["SM3_CONV_V2_DIAG"] = "1"
    ops.conv_gemm(d, x, w, y, None, part)
    os.environ["SM3_CONV_V2_DIAG"] = "0"
