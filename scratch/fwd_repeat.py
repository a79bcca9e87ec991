"""After ONE real fp16 step, repeat the next step with lr = 0 (parameters frozen): is the forward loss reproducible?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
from test_config_gpu import _batch, _build
from sm3hip.trainer import SM3Trainer
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
batches = [_batch(8, 64, 40 + i) for i in range(2)]
m = _build(41, dt)
tr = SM3Trainer(m, lr=1e-4, **({"growth_interval": 1000, "init_scale": 1024.0} if dt == torch.float16 else {}))
print("step 1 loss", float(tr.step(*batches[0])))
tr.lr = 0.0; tr.wd = 0.0
p0 = tr._engine().store.flat_p.clone()
if len(sys.argv) > 2 and sys.argv[2] == "single": tr._engine().two_streams = False
for mode in ({},):
    ls = []
    for r in range(10):
        ls.append(round(float(tr.step(*batches[1])), 6)); torch.cuda.synchronize()
    print(sys.argv[1:], {k: v for k, v in os.environ.items() if k.startswith("SM3_")}, "frozen-parameter repeats of step 2:", ls, "params unchanged:", bool(torch.equal(p0, tr._engine().store.flat_p)))
