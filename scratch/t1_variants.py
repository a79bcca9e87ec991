"""bf16 step at the T1 golden's size (B=4, 64x64) under the linbn switches: which piece moves the loss?"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
    import numpy as np, torch
    from test_e2e_gpu import _build, _batch, _load
    from sm3hip.trainer import SM3Trainer
    out = {}
    for tag in ("b4_s64_f64", "b8_s64_style1_f64", "b32_s64_f64"):
        g = _load(os.path.join(ROOT, "tests", "golden"), tag)
        batch, size, seed, style = [int(v) for v in g["meta"]]
        derm, clinic = _batch(batch, size, seed)
        out[tag + "_golden"] = round(float(g["loss"]), 4)
        for dt in (torch.bfloat16, torch.float16):
            model = _build(seed, dt)
            tr = SM3Trainer(model, lr=1e-3, style=style, init_scale=1024.0) if dt == torch.float16 else SM3Trainer(model, lr=1e-3, style=style)
            loss = float(tr.step(derm, clinic))
            out[f"{tag}_{str(dt)[6:]}"] = round(loss, 4)
    print(json.dumps(out))
else:
    for env in ({}, {"SM3_LINBN": "0"}, {"SM3_LINBN_JOIN": "0"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
