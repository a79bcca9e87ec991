"""Whole-step gradient of the 16-bit modes against the exact-f32 mode of the same engine, under the linbn switches."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd"), os.path.join(ROOT, "tests")]
    import torch
    from test_e2e_gpu import _build
    from test_config_gpu import _latent_batch
    from sm3hip.trainer import SM3Trainer
    out = {}
    for B, S in ((32, 64), (16, 224), (64, 128)):
        derm, clinic = _latent_batch(B, S, 21)
        ref = None
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            model = _build(21, dt)
            tr = SM3Trainer(model, lr=0.0, init_scale=256.0)
            loss = float(tr.step(derm, clinic))
            torch.cuda.synchronize()
            eng = tr._engine()
            g = eng.store.flat_g.double().cpu()
            if dt == torch.float16:
                g = g / float(tr._scaler["scale"]) if tr.steps_taken() == 1 else g * float("nan")
            if ref is None:
                ref = (loss, g)
            else:
                cos = float(torch.dot(g, ref[1]) / (g.norm() * ref[1].norm()))
                out[f"B{B}_S{S}_{str(dt)[6:]}"] = (round(loss - ref[0], 4), round(cos, 3), round(float(g.norm() / ref[1].norm()), 3))
            del tr, model, eng
    print(json.dumps(out))
else:
    for env in ({"SM3_LINBN": "0"}, {"SM3_LINBN_FWD": "0"}, {"SM3_LINBN_JOIN": "0"}, {}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
