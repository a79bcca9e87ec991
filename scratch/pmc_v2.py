"""A few dispatches of both gather-GEMM structures on one MFMA-bound shape, for rocprofv3 --pmc."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops
N = 256
dev = torch.device("cuda:0"); dt, code = torch.bfloat16, 1
Ci, Co, k, s, H = [int(v) for v in os.environ.get("SHAPE", "256,256,3,1,14").split(",")]
d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, k // 2)
x = torch.randn(N, H, H, Ci, device=dev).to(dt)
w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
y = torch.empty(N, d.Ho, d.Wo, Co, dtype=dt, device=dev)
part = torch.empty((N * d.Ho * d.Wo + 127) // 128 * 2 * Co, device=dev)
for v in ("0", "1"):
    os.environ["SM3_CONV_V2"] = v
    for _ in range(3):
        ops.conv_gemm(d, x, w, y, None, part)
    torch.cuda.synchronize()
