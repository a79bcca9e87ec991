"""Minimal inference entry point on the HIP encoders -- mirrors the reference's inference.py surface
(`Extractor`, `MultiLabelProjector`, `Model`, NUM_CLASSES ...; reference inference.py:16-96).

The two ResNet-50 encoders (all of the FLOPs) run on the sm3hip engine in eval mode, and so do the heads
(`sm3hip/heads.py`): the 8 x Linear(4096,512) label projectors as one MFMA GEMM, the 8-token
TransformerEncoderLayer (attention, residual LayerNorms, feed-forward) and the 8 prototype heads on
`csrc/heads.hip`.  The nn.Modules below only own the parameters in the reference's layout; in training mode (the
fine-tuning of tools/mlc_eval.py) the heads run on the train-mode kernels of `sm3hip/mlc.py` / `csrc/heads_train.hip`
(exact-f32 MFMA GEMMs, attention / LayerNorm / dropout / prototype kernels with autograd).  Checkpoints in the reference's wire format
(`best_linear.pth` / `best_finetune.pth`, keys with "encoder." stripped, inference.py:123-127) load unchanged.
"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: see sm3hip/__init__.py

import torch
import torch.nn as nn

import resnet

CLASSES_NAME = ["DIAG", "PN", "BWV", "VS", "PIG", "STR", "DaG", "RS"]
NUM_CLASSES = [5, 3, 2, 3, 3, 3, 3, 2]
CLS_WEIGHTS = [2, 2, 1, 2, 2, 2, 2, 1]
CLASSES_NAME_2 = [f"{n}-{j + 1}" for n, c in zip(CLASSES_NAME, NUM_CLASSES) for j in range(c)]


class MultiLabelProjector(nn.Module):
    """One biased Linear per label (reference inference.py:16-29)."""

    def __init__(self, in_dim, proj_dim, num_labels):
        super().__init__()
        self.projectors = nn.ModuleList([nn.Sequential(nn.Linear(in_dim, proj_dim)) for _ in range(num_labels)])

    def forward(self, x):
        return [p(x) for p in self.projectors]


class Extractor(nn.Module):
    def __init__(self, arch, weights=None) -> None:
        super().__init__()
        self.derm_backbone = resnet.__dict__[arch](weights=weights)
        self.derm_feat_dim = self.derm_backbone.fc.in_features
        self.derm_backbone.fc = nn.Identity()
        self.clinic_backbone = resnet.__dict__[arch](weights=weights)
        self.clinic_feat_dim = self.clinic_backbone.fc.in_features
        self.clinic_backbone.fc = nn.Identity()

    def forward(self):
        pass

    def extract(self, derm_imgs, clinic_imgs):
        return [self.derm_backbone(derm_imgs), self.clinic_backbone(clinic_imgs)]


class Model(nn.Module):
    def __init__(self, extractor, projectors, feat_dim, l2_norm, n_heads, sa_dim_ff, sa_dropout):
        super().__init__()
        self.extractor = extractor
        self.projectors = projectors
        self.mlc_sa = nn.TransformerEncoderLayer(d_model=feat_dim, nhead=n_heads, dim_feedforward=sa_dim_ff,
                                                 dropout=sa_dropout)
        self.feat_dim = feat_dim
        self.l2_norm = l2_norm
        self._hip_heads = None
        self.prototypes = nn.ModuleList([nn.Linear(feat_dim, n) for n in NUM_CLASSES])
        for layer in self.prototypes:
            layer.weight.data.normal_(mean=0.0, std=0.01)
            layer.bias.data.zero_()

    def forward(self, derm_imgs, clinic_imgs):
        feats = torch.cat(self.extractor.extract(derm_imgs, clinic_imgs), dim=1)      # [B, 4096]  (HIP encoders)
        if not self.training and feats.is_cuda:                                       # heads on the HIP kernels
            if self._hip_heads is None:
                from sm3hip.heads import LabelHeads
                self._hip_heads = LabelHeads(self)
            import sm3hip
            dt = self.extractor.derm_backbone.__dict__.get("sm3_dtype") or sm3hip.default_dtype()
            return self._hip_heads(feats.float(), dt)
        if not feats.is_cuda:
            raise RuntimeError("the SM3 HIP path has no CPU fallback: move the model and its inputs to the GPU")
        from sm3hip import mlc                                                        # train mode: heads with autograd
        return mlc.heads_forward(self, feats.float())[1]


def build_model(arch="resnet50", mlc_proj_dim=512, num_labels=8, l2_norm=False, num_heads=1, sa_dim_ff=128,
                sa_dropout=0.1):
    extractor = Extractor(arch)
    feat_dim = extractor.derm_feat_dim + extractor.clinic_feat_dim
    return Model(extractor, MultiLabelProjector(feat_dim, mlc_proj_dim, num_labels), mlc_proj_dim, l2_norm, num_heads,
                 sa_dim_ff, sa_dropout)


if __name__ == "__main__":
    import sys
    pretrain_path = sys.argv[1] if len(sys.argv) > 1 else None
    evaluator = build_model()
    if pretrain_path:
        state_dict = torch.load(pretrain_path, map_location="cpu")["state_dict"]
        for k in list(state_dict.keys()):
            if "encoder." in k:
                state_dict[k.replace("encoder.", "")] = state_dict.pop(k)
        evaluator.load_state_dict(state_dict, strict=True)
        print(f"loaded pre-trained model weights from '{pretrain_path}'")
    evaluator = evaluator.cuda().eval()
    with torch.no_grad():
        preds = evaluator(torch.randn(8, 3, 224, 224).cuda(), torch.randn(8, 3, 224, 224).cuda())
    print([tuple(p.shape) for p in preds])
