import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip import ops, profiler
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = torch.bfloat16; model.to(dev)
tr = SM3Trainer(model, lr=1e-6)
tr._engine().two_streams = False
g = torch.Generator(device=dev).manual_seed(1)
derm = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
clinic = [torch.randn(B, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
for _ in range(2): tr.step(derm, clinic)
torch.cuda.synchronize()
# host enqueue time vs GPU time
t0 = time.perf_counter(); tr.step(derm, clinic); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0):.1f} ms, step wall {1e3*(t2-t0):.1f} ms")
p = profiler.Profiler(detail=True)
ops.set_profiler(p); tr.step(derm, clinic); torch.cuda.synchronize(); ops.set_profiler(None)
print(p.format_table(f"detail B={B}"))
