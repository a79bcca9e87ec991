"""Is the fp16 4-step run of tests/test_round3_gpu.py::test_fp16_resume_... deterministic?  Repeats it R times per
environment and prints per-step loss, scale after the step and steps taken."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
from test_config_gpu import _batch, _build
from sm3hip.trainer import SM3Trainer
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batches = [_batch(8, 64, 40 + i) for i in range(4)]
envs = json.loads(os.environ.get("VARIANTS", '[{}]'))
keys = sorted({k for v in envs for k in v})
for env in envs:
    for k in keys:
        if k in env: os.environ[k] = env[k]
        else: os.environ.pop(k, None)
    seen = {}
    for r in range(R):
        m = _build(41, torch.float16)
        tr = SM3Trainer(m, lr=1e-4, growth_interval=3, init_scale=1024.0)
        rec = []
        for i in range(4):
            loss = tr.step(*batches[i]); torch.cuda.synchronize()
            sc = tr.scaler_state_dict()
            rec.append((round(float(loss), 6), sc["scale"], sc["_growth_tracker"], tr.steps_taken()))
        seen[str(rec)] = seen.get(str(rec), 0) + 1
    print(json.dumps(env), flush=True)
    for k, n in sorted(seen.items(), key=lambda kv: -kv[1]): print(f"   {n:3d} x {k}", flush=True)
