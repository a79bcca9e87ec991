"""Is the SECOND trainer created in a process slower than the first (same dtype, same inputs)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
import torch
from sm3hip.trainer import SM3Trainer
from src.models.simclr import SimCLRSkinV32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
d = [torch.randn(256, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
c = [torch.randn(256, 3, 224, 224, device=dev, generator=g) for _ in range(2)]
seq = sys.argv[1:] or ["f16", "f16", "f16"]
for i, name in enumerate(seq):
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[name]
    torch.manual_seed(3407)
    m = SimCLRSkinV32("resnet50", None, 128, 0.1); m.sm3_dtype = dt; m.to(dev)
    tr = SM3Trainer(m, lr=1e-6, weight_decay=5e-2, eps=1e-5, style=0)
    for _ in range(4):
        tr.step(d, c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        tr.step(d, c)
    torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
    eng = tr._engine()
    print(f"  flat_p at {eng.store.flat_p.data_ptr():#x}, lane streams {[st.cuda_stream for st in (eng._streams or {}).values()]}, "
          f"side {[st.cuda_stream for st in eng._side.values()]}")
    print(f"trainer {i} ({name}): {256 * 10 / dt_:.0f} pairs/s, streams in engine: {len(tr._engine()._streams or {})}", flush=True)
    if os.environ.get("KEEP") != "1":
        del tr, m
        if os.environ.get("NO_EMPTY") != "1":
            torch.cuda.empty_cache()
