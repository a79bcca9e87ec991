import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops
D = torch.device("cuda:0"); dt = torch.bfloat16; code = ops.dtype_code(dt)
N, H, Ci = 3, 14, 256
d = ops.fwd_desc(code, N, H, H, Ci, Ci, 3, 1, 1); M = N * H * H
g = torch.Generator().manual_seed(1)
x = torch.randn(M, Ci, generator=g).to(dt).to(D); w = (torch.randn(Ci, 9 * Ci, generator=g) / 48).to(dt).to(D)
ys = []
for mode in ("0", "1"):
    os.environ["SM3_CONV_HALO"] = mode
    y = torch.empty(M, Ci, dtype=dt, device=D); ops.conv_gemm(d, x, w, y, None, None); ys.append(y)
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x.float().view(N, H, H, Ci).permute(0, 3, 1, 2).cpu().double(), w.float().view(Ci, 3, 3, Ci).permute(0, 3, 1, 2).cpu().double(), padding=1).permute(0, 2, 3, 1).reshape(M, Ci)
for y in ys: print("max err vs f64", float((y.cpu().double() - ref).abs().max()), "equal to each other", bool(torch.equal(ys[0], ys[1])))
