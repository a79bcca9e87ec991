import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[os.environ.get("DT", "bf16")]
torch.manual_seed(3407)
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = dt; model.to(dev)
tr = SM3Trainer(model, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0, data_parallel=True)
g = torch.Generator(device=dev).manual_seed(3407)
B, S = 32, 64
derm = [torch.randn(B, 3, S, S, device=dev, generator=g) for _ in range(2)]
clinic = [torch.randn(B, 3, S, S, device=dev, generator=g) for _ in range(2)]
def run(tag, n=3):
    print(tag, [round(float(tr.step(derm, clinic)), 4) for _ in range(n)], flush=True)
run("rccl")
os.environ["SM3_SYNCBN_P2P"] = "1"; tr._engine().__dict__["_explicit_sync"] = None; tr._engine()
run("p2p ")
tr.check(); tr.close(); os.environ["SM3_SYNCBN_P2P"] = "0"; tr._engine()
run("rccl")
tr2 = SM3Trainer(model, lr=1e-6, data_parallel=False)
print("single-rank trainer on the same model:", [round(float(tr2.step(derm, clinic)), 4) for _ in range(2)])
dist.destroy_process_group()
