import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
dev = torch.device("cuda:0")
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = torch.bfloat16; model.to(dev)
tr = SM3Trainer(model, lr=1e-6)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
derm = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)]
clinic = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)]
for _ in range(3): tr.step(derm, clinic)
torch.cuda.synchronize()
print("B", B, "peak allocated GB", torch.cuda.max_memory_allocated() / 1e9, "reserved GB", torch.cuda.max_memory_reserved() / 1e9)
