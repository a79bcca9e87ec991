"""Where the host time of one step goes (cProfile, batch 32 so the GPU is not the bottleneck)."""
import cProfile, pstats, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skin-sm3_amd"))
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
dev = torch.device("cuda:0")
model = SimCLRSkinV32("resnet50", None, 128, 0.1); model.sm3_dtype = torch.bfloat16; model.to(dev)
tr = SM3Trainer(model, lr=1e-6)
B = 32
derm = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)]
clinic = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)]
for _ in range(3): tr.step(derm, clinic)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.step(derm, clinic)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
