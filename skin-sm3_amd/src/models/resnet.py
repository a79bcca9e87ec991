"""ResNet encoders on the SM3 HIP engine -- drop-in for the reference's src/models/resnet.py.

Same public surface as the reference file (resnet.py:177-329, 724-751): `ResNet`, `Bottleneck`,
`resnet50/101/152(weights=None, progress=True, **kwargs)`, attribute names (`conv1, bn1, relu, maxpool,
layer1..4, avgpool, fc`) and therefore the same state_dict keys, Kaiming fan_out initialisation
(resnet.py:227-232) and `zero_init_residual`.  The nn.Conv2d / nn.BatchNorm2d children are parameter
containers only: `forward` runs the hand-written gfx950 kernels through sm3hip.engine (stem im2col +
MFMA GEMM, gather-GEMM convolutions with BN-statistics epilogues, fused BN/residual/ReLU, pooling).
No torchvision dependency.  BasicBlock architectures (resnet18/34) are not on the SM3 path
(run.sh uses resnet50 only) and are rejected.
"""
import os
from typing import Any, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

__all__ = ["ResNet", "Bottleneck", "resnet50", "resnet101", "resnet152"]


def conv3x3(in_planes: int, out_planes: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes: int, out_planes: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    """Parameter container for one ResNet-v1.5 bottleneck (stride on the 3x3, reference resnet.py:119-174)."""
    expansion: int = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None,
                 norm_layer=None) -> None:
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = norm_layer(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = norm_layer(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:  # pragma: no cover - blocks are sequenced by the engine
        raise RuntimeError("Bottleneck blocks are executed by sm3hip.engine, not called individually")


class ResNet(nn.Module):
    def __init__(self, block, layers: List[int], num_classes: int = 1000, zero_init_residual: bool = False,
                 norm_layer=None) -> None:
        super().__init__()
        if block is not Bottleneck:
            raise NotImplementedError("the SM3 HIP engine implements Bottleneck ResNets (resnet50/101/152)")
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.block_counts = list(layers)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck) and m.bn3.weight is not None:
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes: int, blocks: int, stride: int = 1) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       self._norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self._norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=self._norm_layer))
        return nn.Sequential(*layers)

    # ---- execution on the HIP engine ------------------------------------------------------
    def _engine(self):
        from sm3hip.bridge import encoder_engine_for
        return encoder_engine_for(self)

    def _forward_impl(self, x: Tensor) -> Tensor:
        from sm3hip.bridge import encoder_features
        feat = encoder_features(self, x)  # [N, 2048] fp32: conv1 ... avgpool + flatten
        return self.fc(feat)

    def forward(self, x: Tensor) -> Tensor:
        return self._forward_impl(x)


def _load_weights(model: ResNet, weights: Any, hub_file: str) -> None:
    """`weights` may be None, a path to a state_dict file, or a torchvision enum name such as
    "IMAGENET1K_V1" (tools/backbone_train.py passes args.arch_weights); the latter is served from the
    local torch-hub cache only -- this build never downloads."""
    if weights is None:
        return
    path = str(weights)
    if not os.path.isfile(path):
        cache = os.path.join(torch.hub.get_dir(), "checkpoints", hub_file)
        if not os.path.isfile(cache):
            raise RuntimeError(f"weights={weights!r}: no local file and no cached {cache}; "
                               "pass weights=None or a state_dict path (no network access)")
        path = cache
    state = torch.load(path, map_location="cpu")
    model.load_state_dict(state.get("state_dict", state))


def _resnet(layers: List[int], weights: Any, hub_file: str, **kwargs: Any) -> ResNet:
    kwargs.pop("progress", None)
    model = ResNet(Bottleneck, layers, **kwargs)
    _load_weights(model, weights, hub_file)
    return model


def resnet50(*, weights: Any = None, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet([3, 4, 6, 3], weights, "resnet50-0676ba61.pth", **kwargs)


def resnet101(*, weights: Any = None, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet([3, 4, 23, 3], weights, "resnet101-63fe2227.pth", **kwargs)


def resnet152(*, weights: Any = None, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet([3, 8, 36, 3], weights, "resnet152-394f9c45.pth", **kwargs)
