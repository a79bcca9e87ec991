"""Glue between the nn.Module mirrors (src/models) and the HIP engine: engine cache, the torch.autograd
bridge used by the drop-in `model(derm_imgs, clinic_imgs, style)` call contract, and SyncBatchNorm /
data-parallel detection.  Autograd is plumbing here: one Function spans the whole hot path; its backward
runs our kernels and hands PyTorch finished parameter gradients (so DDP hooks, torch optimizers and
GradScaler keep working unchanged)."""
import torch
import torch.distributed as dist
import torch.nn as nn

import sm3hip
from . import ops
from .engine import SM3Engine


def _engine_for(module, kind):
    eng = module.__dict__.get("_sm3_engine")
    dtype = module.__dict__.get("sm3_dtype") or sm3hip.default_dtype()
    if eng is None or eng.tdt != dtype or eng.kind != kind:
        eng = SM3Engine(module, dtype, kind)
        module.__dict__["_sm3_engine"] = eng  # not a submodule / buffer: invisible to state_dict and .to()
    _configure_sync(module, eng)
    return eng


def _configure_sync(module, eng):
    """SyncBatchNorm.convert_sync_batchnorm(model) (tools/backbone_train.py:510) swaps the BN containers for
    nn.SyncBatchNorm: when that happened and a process group is up, BN statistics are summed over ranks."""
    has_sync = any(isinstance(m, nn.SyncBatchNorm) for m in module.modules())
    if has_sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if eng.stat_sync is None:
            eng.world_size = dist.get_world_size()
            eng.stat_sync = lambda t: dist.all_reduce(t)
    elif eng.__dict__.get("_explicit_sync") is None:
        eng.stat_sync, eng.world_size = None, 1


def encoder_engine_for(resnet):
    return _engine_for(resnet, "encoder")


def sm3_engine_for(model, kind):
    return _engine_for(model, kind)


def _needs_grad(module):
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


def _params(module):
    return [p for _, p in module.named_parameters()]


def _scratch_backward(eng, fn):
    """Run fn() with the engine's flat gradient buffer swapped for a zeroed scratch buffer; return per-parameter
    gradient views of the scratch (what autograd accumulates into .grad)."""
    scratch = torch.zeros_like(eng.store.flat_g)
    old = eng.store.flat_g
    eng.store.flat_g = scratch
    try:
        fn()
        eng._sync_side()  # weight-gradient kernels run on a side stream
    finally:
        eng.store.flat_g = old
    return eng.store.grad_views(scratch)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, branch, train, x, *params):
        feat, ectx = eng.encoder_only(branch, x, train, True)
        ctx.eng, ctx.ectx = eng, ectx
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eng = ctx.eng
        d = torch.empty(dfeat.shape, dtype=eng.tdt, device=dfeat.device)
        ops.cast_from_f32(eng.dtype, dfeat.contiguous().float(), d)
        grads = _scratch_backward(eng, lambda: eng.encoder_backward(ctx.ectx, d))
        ctx.ectx = None
        return (None, None, None, None) + tuple(grads)


def encoder_features(resnet, x):
    """conv1 ... avgpool + flatten of a bare ResNet on the HIP engine: [N,2048] fp32."""
    eng = encoder_engine_for(resnet)
    if _needs_grad(resnet):
        return _EncoderFn.apply(eng, "main", resnet.training, x, *_params(resnet))
    feat, _ = eng.encoder_only("main", x, resnet.training, False)
    return feat


def branch_features(model, kind, branch, x):
    """SimCLRSkinV3.extract: one encoder of the SM3 model, run by the model's own engine (same dtype, same
    flat parameter store); train/eval statistics follow that encoder module's flag, as nn.Module would."""
    eng = sm3_engine_for(model, kind)
    enc = getattr(model, branch + "_backbone").encoder
    if _needs_grad(enc):
        return _EncoderFn.apply(eng, branch, enc.training, x, *_params(model))
    feat, _ = eng.encoder_only(branch, x, enc.training, False)
    return feat


class _ModelFn(torch.autograd.Function):
    """forward: images -> the reference-layout NT-Xent logits of every loss term;
    backward: d(logits) -> parameter gradients."""

    @staticmethod
    def forward(ctx, eng, views, style, train, temperature, *params):
        zs, _feats, saved = eng.forward(views, style, train, True)
        eng.last_feats = {k: v[0] for k, v in _feats.items()}  # fp32 pooled features of this very forward (detached)
        outs, nt = [], []
        for name, z in zs.items():
            R = z.shape[0]
            zn, inv = torch.empty_like(z), torch.empty(R, dtype=torch.float32, device=z.device)
            logits = torch.empty(R, R - 1, dtype=torch.float32, device=z.device)
            ops.ntxent_logits(z, temperature, zn, inv, logits)
            outs.append(logits)
            nt.append((name, zn, inv))
        ctx.eng, ctx.saved, ctx.nt, ctx.temperature = eng, saved, nt, temperature
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dlogits):
        eng = ctx.eng
        dz = {}
        for (name, zn, inv), dl in zip(ctx.nt, dlogits):
            out = torch.empty(zn.shape, dtype=eng.tdt, device=zn.device)
            if dl is None:
                out.zero_()
            else:
                ops.ntxent_logits_bwd(eng.dtype, dl.contiguous().float(), zn, inv, ctx.temperature, out)
            dz[name] = out
        grads = _scratch_backward(eng, lambda: eng.backward(ctx.saved, dz))
        ctx.saved = None
        return (None, None, None, None, None) + tuple(grads)


def model_logits(model, kind, views, style, temperature):
    """List of logits tensors [2B, 2B-1] (fp32) in the order: branches..., cross pairs...  With grad mode on
    they are connected to the parameters through _ModelFn."""
    eng = sm3_engine_for(model, kind)
    if _needs_grad(model):
        return list(_ModelFn.apply(eng, views, style, model.training, temperature, *_params(model)))
    zs, _feats, _ = eng.forward(views, style, model.training, False)
    eng.last_feats = {k: v[0] for k, v in _feats.items()}
    outs = []
    for z in zs.values():
        R = z.shape[0]
        zn, inv = torch.empty_like(z), torch.empty(R, dtype=torch.float32, device=z.device)
        logits = torch.empty(R, R - 1, dtype=torch.float32, device=z.device)
        ops.ntxent_logits(z, temperature, zn, inv, logits)
        outs.append(logits)
    return outs
