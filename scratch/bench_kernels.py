"""Micro-benchmark of the MFMA kernels over the ResNet-50 layer shapes at batch N (bf16), HIP-event timed,
variants interleaved in one process (cdna guide rule 24)."""
import os, sys, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops
N = int(os.environ.get("NB", "256"))
what = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
dev = torch.device("cuda:0")
dt, code = torch.bfloat16, 1
# (Ci, Co, k, s, H) distinct ResNet-50 convs with multiplicity
SHAPES = [(64,64,1,1,56,1),(64,64,3,1,56,3),(64,256,1,1,56,4),(256,64,1,1,56,2),(256,128,1,1,56,1),(128,128,3,2,56,1),
          (128,512,1,1,28,4),(256,512,1,2,56,1),(512,128,1,1,28,3),(128,128,3,1,28,3),(512,256,1,1,28,1),(256,256,3,2,28,1),
          (256,1024,1,1,14,6),(512,1024,1,2,28,1),(1024,256,1,1,14,5),(256,256,3,1,14,5),(1024,512,1,1,14,1),(512,512,3,2,14,1),
          (512,2048,1,1,7,3),(1024,2048,1,2,14,1),(2048,512,1,1,7,2),(512,512,3,1,7,2)]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
variants = {
  "wgrad": [{"SM3_WGRAD_KP": "32", "SM3_WGRAD_TARGET_CTAS": "1024"}, {"SM3_WGRAD_KP": "32", "SM3_WGRAD_TARGET_CTAS": "512"},
            {"SM3_WGRAD_KP": "32", "SM3_WGRAD_TARGET_CTAS": "256"}, {"SM3_WGRAD_KP": "64", "SM3_WGRAD_TARGET_CTAS": "512"},
            {"SM3_WGRAD_KP": "64", "SM3_WGRAD_TARGET_CTAS": "256"}],
  "fwd": [dict(v) for v in eval(os.environ.get("FWD_VARIANTS", "[{}]"))],
}[what]
tot = [0.0] * len(variants)
print("shape".ljust(34), *[str(sorted(v.items())) for v in variants])
for (Ci, Co, k, s, H, mult) in SHAPES:
    p = k // 2
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, s, p)
    x = torch.randn(N, H, H, Ci, device=dev).to(dt)
    w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
    y = torch.empty(N, d.Ho, d.Wo, Co, dtype=dt, device=dev)
    dy = torch.randn(N, d.Ho, d.Wo, Co, device=dev).to(dt)
    dw = torch.zeros(Co, k * k * Ci, device=dev)
    part = torch.empty((N * d.Ho * d.Wo + 127) // 128 * 2 * Co, device=dev)
    flops = 2.0 * N * d.Ho * d.Wo * Co * k * k * Ci
    row = []
    for vi, v in enumerate(variants):
        os.environ.update(v)
        if what == "wgrad":
            t = timeit(lambda: ops.conv_wgrad(d, x, dy, dw))
        else:
            t = timeit(lambda: ops.conv_gemm(d, x, w, y, None, part))
        tot[vi] += t * mult
        row.append(f"{t*1e3:8.1f}us {flops/t/1e9:6.0f}TF")
    print(f"Ci{Ci} Co{Co} k{k} s{s} H{H} x{mult}".ljust(34), *row)
print("weighted total ms (x4 encoder passes):", [round(4 * t, 2) for t in tot])
