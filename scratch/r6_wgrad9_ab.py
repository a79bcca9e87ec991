"""Nine-tap owner weight gradient vs the tap-shifted kernel on the step's four 3x3 geometries (B = 256: N = 512 images per
launch), deterministic (slab) form as the engine calls it; ms per launch and TFLOP/s, interleaved rounds.
usage: python scratch/r6_wgrad9_ab.py [bf16|f16]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "skin-sm3_amd"))
import torch
from sm3hip import ops

dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
code = ops.dtype_code(dt)
dev = torch.device("cuda:0")
SHAPES = [(512, 64, 64, 56, 56), (512, 128, 128, 28, 28), (512, 256, 256, 14, 14), (512, 512, 512, 7, 7)]
def v9(r, n):
    return (f"nine-tap R={r} NST={n}", {"SM3_WGRAD9": "1", "SM3_WGRAD9_R": str(r), "SM3_WGRAD9_NST": str(n)})


VARIANTS = [("tap-shifted", {"SM3_WGRAD9": "0"}), ("nine-tap (default rule)", {"SM3_WGRAD9": "1"})]
EXTRA = {  # per-shape experiments: rows per stage, ring depth
    (512, 64, 64, 56, 56): [v9(1, 2), v9(1, 1), v9(2, 1)],
    (512, 128, 128, 28, 28): [v9(4, 2), v9(4, 1), v9(7, 1), v9(2, 2)],
    (512, 256, 256, 14, 14): [v9(7, 2), v9(7, 1), v9(14, 1), v9(2, 2)],
    (512, 512, 512, 7, 7): [v9(7, 2), v9(7, 1)],
}
KEYS = ["SM3_WGRAD9", "SM3_WGRAD9_R", "SM3_WGRAD9_NST"]


def run(d, x, dy, dw, slabs, cap, env, reps):
    for k in KEYS:
        os.environ.pop(k, None)
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


for shp in SHAPES:
    N, Ci, Co, H, W = shp
    x = torch.randn(N, H, W, Ci, device=dev).to(dt)
    dy = torch.randn(N, H, W, Co, device=dev).to(dt)
    d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
    n = Co * 9 * Ci
    cap = int(os.environ.get('SM3_EXP_CAP', 0)) or ops.wgrad_det_cap(n)
    slabs = torch.empty(cap * n, device=dev)
    dw = torch.zeros(Co, 9 * Ci, device=dev)
    fl = 2.0 * N * H * W * Co * 9 * Ci
    vs = VARIANTS + EXTRA.get(shp, [])
    best = {name: 1e9 for name, _ in vs}
    for rnd_ in range(3):
        for name, env in vs:
            best[name] = min(best[name], run(d, x, dy, dw, slabs, cap, env, 10))
    print(f"M{N*H*W}_K9x{Ci}_N{Co} (H=W={H}):")
    for name, _ in vs:
        print(f"   {name:24s} {best[name]*1e3:8.1f} us  {fl/best[name]/1e9:7.1f} TFLOP/s")
