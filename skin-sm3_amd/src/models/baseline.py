"""Dual-backbone linear-probe / fine-tune model on the HIP encoders -- drop-in for the reference's
`src/models/baseline.py:60-117` (`Baseline`), the model of `tools/backbone_eval.py`.

Two ResNet encoders (fc = Identity) run on sm3hip; under `--finetune fc` they are frozen and in eval mode, where
every conv + BatchNorm (+residual) (+ReLU) is ONE kernel (sm3_conv_bn_act_eval).  The eight classification
heads (Linear(4096, n_i), 98 K MACs per sample) stay stock PyTorch, as SURVEY.md 2.1 #5 scopes them.
`SingleBaseline` / `BaselineMLP1-3` of the reference file have no caller and are not provided; timm backbones
are not on the SM3 path.
"""
import torch
import torch.nn as nn

from .resnet import resnet50, resnet101, resnet152

NUM_CLASSES = [5, 3, 2, 3, 3, 3, 3, 2]  # DIAG, PN, BWV, VS, PIG, STR, DaG, RS (backbone_eval.py:60-62)


class Baseline(nn.Module):
    def __init__(self, arch="resnet50", weights=None):
        super().__init__()
        ctor = {"resnet50": resnet50, "resnet101": resnet101, "resnet152": resnet152}.get(arch)
        if ctor is None:
            raise NotImplementedError(f"arch {arch!r}: the SM3 HIP engine implements Bottleneck ResNets")
        self.derm_backbone = ctor(weights=weights)
        self.clinic_backbone = ctor(weights=weights)
        feat_dim = 2048 * 2
        self.derm_backbone.fc = nn.Identity()
        self.clinic_backbone.fc = nn.Identity()
        self.classifier = nn.ModuleList([nn.Linear(feat_dim, n) for n in NUM_CLASSES])
        for layer in self.classifier:
            layer.weight.data.normal_(mean=0.0, std=0.01)
            layer.bias.data.zero_()

    def forward(self, x):
        derm_feats = self.derm_backbone(x[0])
        clinic_feats = self.clinic_backbone(x[1])
        feats = torch.cat([derm_feats, clinic_feats], dim=1)
        return [classify(feats) for classify in self.classifier]

    def freeze_backbone(self):
        for p in list(self.derm_backbone.parameters()) + list(self.clinic_backbone.parameters()):
            p.requires_grad = False

    def unfreeze_backbone(self):
        for p in list(self.derm_backbone.parameters()) + list(self.clinic_backbone.parameters()):
            p.requires_grad = True
