"""Which aten ops does one step issue in the forced data-parallel mode?  (torch.profiler, CPU activities only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
import torch, torch.distributed as dist
dp = len(sys.argv) > 1 and sys.argv[1] == "dp"
torch.cuda.set_device(0); dev = torch.device("cuda:0")
if dp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
torch.manual_seed(0)
m = SimCLRSkinV32("resnet50", None, 128, 0.1); m.sm3_dtype = torch.bfloat16; m.to(dev)
tr = SM3Trainer(m, lr=1e-6, data_parallel=True if dp else None)
g = torch.Generator(device=dev).manual_seed(1)
d = [torch.randn(64, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
c = [torch.randn(64, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
for _ in range(2):
    tr.step(d, c)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(d, c)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)[:14]
for e in rows:
    print(f"{e.key[:50]:50s} {e.count:6d}")
ka = prof.key_averages(group_by_stack_n=6)
for e in sorted(ka, key=lambda e: -e.count)[:6]:
    if e.key in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::zeros", "aten::empty", "aten::slice"):
        print("\n", e.key, e.count)
        for s in e.stack[:6]:
            print("    ", s)
if dp:
    dist.destroy_process_group()
