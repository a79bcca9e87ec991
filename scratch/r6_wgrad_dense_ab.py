"""Dense (1x1) weight gradient of the layer-3 / layer-4 shapes at B = 256 (N = 512 images per launch), deterministic form:
the shipping ring (32-pixel steps, 4 stages x 2 K-groups, one workgroup per CU) against one-stage variants with large stages
(SM3_WGRAD_DENSE_KP = 64 / 128: 2 workgroups per CU, the structure of the nine-tap owner).  ms per launch, TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "skin-sm3_amd"))
import torch
from sm3hip import ops

dt = torch.bfloat16
code = ops.dtype_code(dt)
dev = torch.device("cuda:0")
SHAPES = [(512, 1024, 256, 14), (512, 256, 1024, 14), (512, 2048, 512, 7), (512, 512, 2048, 7), (512, 512, 512, 7), (512, 256, 256, 14),
          (512, 512, 128, 28)]
VARIANTS = [("ring 32 x 4 x 2 (shipping)", {}), ("one stage of 64", {"SM3_WGRAD_DENSE_KP": "64"}),
            ("one stage of 128", {"SM3_WGRAD_DENSE_KP": "128"})]


def run(d, x, dy, dw, slabs, cap, env, reps):
    os.environ.pop("SM3_WGRAD_DENSE_KP", None)
    os.environ.update(env)
    ops.conv_wgrad_det(d, x, dy, dw, slabs, cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_wgrad_det(d, x, dy, dw, slabs, cap)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for N, Ci, Co, H in SHAPES:
    x = torch.randn(N, H, H, Ci, device=dev).to(dt)
    dy = torch.randn(N, H, H, Co, device=dev).to(dt)
    d = ops.fwd_desc(code, N, H, H, Ci, Co, 1, 1, 0)
    n = Co * Ci
    cap = ops.wgrad_det_cap(n)
    slabs = torch.empty(cap * n, device=dev)
    dw = torch.zeros(Co, Ci, device=dev)
    fl = 2.0 * N * H * H * Co * Ci
    best = {name: 1e9 for name, _ in VARIANTS}
    ref = None
    for r in range(3):
        for name, env in VARIANTS:
            best[name] = min(best[name], run(d, x, dy, dw, slabs, cap, env, 10))
    print(f"M{N*H*H}_K1x{Ci}_N{Co}:")
    for name, _ in VARIANTS:
        print(f"   {name:28s} {best[name]*1e3:8.1f} us  {fl/best[name]/1e9:7.1f} TFLOP/s")
