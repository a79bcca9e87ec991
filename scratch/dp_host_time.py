"""Host enqueue time vs GPU time of one step: plain, and with the data-parallel path forced on one rank (nccl backend; with and
without SM3_SYNCBN_P2P=1).  Usage: python3 scratch/dp_host_time.py [dp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
import torch
import torch.distributed as dist
dp = len(sys.argv) > 1 and sys.argv[1] == "dp"
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
if dp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
torch.manual_seed(0)
m = SimCLRSkinV32("resnet50", None, 128, 0.1); m.sm3_dtype = torch.bfloat16; m.to(dev)
tr = SM3Trainer(m, lr=1e-6, data_parallel=True if dp else None)
if len(sys.argv) > 1 and sys.argv[1] == "fake":  # the data-parallel CODE PATH of the engine with an exchange that does nothing
    eng = tr._engine()
    eng.stat_sync = lambda t: None
    eng.world_size = 1
if len(sys.argv) > 1 and sys.argv[1] in ("fake", "notify") and os.environ.get("FAKE_NOTIFY", "1") == "1":
    eng = tr._engine()
    _bw = eng.backward

    def bw(saved, dz, dfeat=None):  # gradient-ready notifications as the data-parallel trainer installs them, doing nothing
        eng.grad_ready = lambda f, l: None
        try:
            return _bw(saved, dz, dfeat)
        finally:
            eng.grad_ready = None
    eng.backward = bw
if os.environ.get("NO_BUCKETS") == "1":  # statistics exchanges stay, the gradient buckets' all-reduces are skipped
    tr._bucket_ready = lambda *a: None
g = torch.Generator(device=dev).manual_seed(1)
d = [torch.randn(256, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
c = [torch.randn(256, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
for _ in range(3):
    tr.step(d, c)
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(5):
    t0 = time.perf_counter(); tr.step(d, c); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); ts.append(t2 - t0)
print(f"mode={sys.argv[1] if len(sys.argv) > 1 else 'plain'} dp={dp} p2p={os.environ.get('SM3_SYNCBN_P2P', '0')}: host enqueue {1e3 * min(hs):.1f} ms, step {1e3 * min(ts):.1f} ms")
if dp:
    dist.destroy_process_group()
