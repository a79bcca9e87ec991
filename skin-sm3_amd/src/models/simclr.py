"""SM3 / SimCLR models on the HIP engine -- drop-in for the reference's src/models/simclr.py.

Public surface kept from the reference: `make_projector(in_dim, proj_dim)` (simclr.py:17-27),
`SimCLR(arch, weights, proj_dim, temperature, return_feats)` (:31-96), `SimCLRSkinV3(...)` (:250-396) and
`SimCLRSkinV32(...)` (:399-482) with the same constructor arguments, attribute names
(`derm_backbone`, `clinic_backbone`, `cross_proj`, `.encoder`, `.projector`, `.encoder_out_dim`,
`derm_feat_dim`, ...), state_dict keys, call contract `model(derm_imgs, clinic_imgs, style)` ->
`((logits, labels), (logits, labels), ((logits, labels), ...))` and `extract(derm, clinic)`.
The children are parameter containers; the arithmetic runs in sm3hip (gfx950 kernels).  The unused
variants SimCLRSkin / V2 / V21 / V22 / V23 (never constructed by any tool) are not provided.
"""
import torch
import torch.nn as nn

from src.models import resnet


def make_projector(in_dim, proj_dim):
    return nn.Sequential(
        nn.Linear(in_dim, in_dim, bias=False),
        nn.BatchNorm1d(in_dim),
        nn.ReLU(inplace=True),
        nn.Linear(in_dim, in_dim, bias=False),
        nn.BatchNorm1d(in_dim),
        nn.ReLU(inplace=True),
        nn.Linear(in_dim, proj_dim, bias=False),
        nn.BatchNorm1d(proj_dim, affine=False),
    )


def _labels(logits):
    return torch.zeros(logits.shape[0], dtype=torch.long, device=logits.device)


class SimCLR(nn.Module):
    def __init__(self, arch, weights=None, proj_dim=128, temperature=0.5, return_feats=False):
        super().__init__()
        self.proj_dim = proj_dim
        self.temperature = temperature
        self.return_feats = return_feats
        self.encoder = resnet.__dict__[arch](weights=weights)
        self.encoder_out_dim = self.encoder.fc.in_features
        self.encoder.fc = nn.Identity()
        self.projector = make_projector(self.encoder_out_dim, self.proj_dim)

    def forward(self, x1, x2):
        from sm3hip import bridge
        if self.return_feats:
            # The pooled features of THIS forward (reference simclr.py:58-59,90: f1, f2 = encoder(x1), encoder(x2) are
            # the very tensors the projector saw) -- no second pass, so BatchNorm running statistics and
            # num_batches_tracked advance exactly as in the reference.  Returned detached: features with autograd attached
            # are only produced inside SimCLRSkinV3 / V32, where they feed the cross projectors of the same fused graph.
            (logits,) = bridge.model_logits(self, "simclr", {"main": [x1, x2]}, 0, self.temperature)
            f = bridge.sm3_engine_for(self, "simclr").last_feats["main"].detach()
            B = x1.shape[0]
            return (logits, _labels(logits)), (f[:B], f[B:])
        (logits,) = bridge.model_logits(self, "simclr", {"main": [x1, x2]}, 0, self.temperature)
        return (logits, _labels(logits))

    def extract(self, imgs):
        return self.encoder(imgs)


class SimCLRSkinV3(nn.Module):
    """Contrast derm vs clinic features with one shared cross projector (reference simclr.py:250-396)."""
    _KIND = "v3"

    def __init__(self, arch, weights=None, proj_dim=128, temperature=0.5, use_checkpoint=False) -> None:
        super().__init__()
        self.temperature = temperature
        self.proj_dim = proj_dim
        self.derm_backbone = SimCLR(arch, weights, proj_dim, temperature, True)
        self.clinic_backbone = SimCLR(arch, weights, proj_dim, temperature, True)
        self.derm_feat_dim = self.derm_backbone.encoder_out_dim
        self.clinic_feat_dim = self.clinic_backbone.encoder_out_dim
        self.cross_feat_dim = self.derm_feat_dim
        self.cross_proj = make_projector(self.cross_feat_dim, proj_dim)
        # use_checkpoint (reference _apply_checkpoint, simclr.py:269-288) trades recompute for memory on
        # 16-80 GB parts; with 288 GB of HBM3E the engine keeps activations resident, so the flag is accepted
        # and ignored (numerically a no-op in the reference too).
        self.use_checkpoint = use_checkpoint

    def forward(self, derm_imgs, clinic_imgs, style):
        from sm3hip import bridge
        if style not in (0, 1, 2):
            raise ValueError("style must be 0, 1 or 2")
        outs = bridge.model_logits(self, self._KIND, {"derm": list(derm_imgs), "clinic": list(clinic_imgs)}, style,
                                   self.temperature)
        pairs = [(lg, _labels(lg)) for lg in outs]
        return (pairs[0], pairs[1], tuple(pairs[2:]))

    def extract(self, derm_imgs, clinic_imgs):
        from sm3hip import bridge
        return [bridge.branch_features(self, self._KIND, "derm", derm_imgs),
                bridge.branch_features(self, self._KIND, "clinic", clinic_imgs)]


METADATA_PAD = 64  # the metadata vector is zero-padded to one 128-byte K chunk of the MFMA GEMM


class SimCLRSkinV32(SimCLRSkinV3):
    """Independent cross projectors per modality (reference simclr.py:399-482; run.sh:4 uses this one).

    metadata_dim (extension, default None = the reference's model, identical state_dict): BASELINE.json's north_star
    names a "metadata-MLP branch" over a 20-dim metadata vector; the reference has none (its metadata columns only become
    label dicts).  With metadata_dim = d <= 64 the model owns `meta_proj`, a BN-MLP projector
    (make_projector over the zero-padded vector), whose output is contrasted with the cross-modal projections of the
    first views by two more NT-Xent terms in sm3hip.trainer.SM3Trainer.step(..., metadata=)."""
    _KIND = "v32"

    def __init__(self, arch, weights=None, proj_dim=128, temperature=0.5, use_checkpoint=False, metadata_dim=None) -> None:
        super().__init__(arch, weights, proj_dim, temperature)
        self.cross_proj = nn.ModuleList([make_projector(self.derm_feat_dim, proj_dim),
                                         make_projector(self.clinic_feat_dim, proj_dim)])
        self.use_checkpoint = use_checkpoint
        self.metadata_dim = metadata_dim
        if metadata_dim is not None:
            if not 0 < metadata_dim <= METADATA_PAD:
                raise ValueError(f"metadata_dim must be in 1..{METADATA_PAD}")
            self.meta_proj = make_projector(METADATA_PAD, proj_dim)
