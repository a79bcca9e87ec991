"""f-4: GPU-side SimCLR augmentation (csrc/augment.hip + sm3hip/augment.py) against the CPU restatement of the reference's
torchvision / PIL chain (oracle/augment_oracle.py; tools/backbone_train.py:448-466)."""
import math

import pytest
import torch

from oracle import augment_oracle as A

MEAN, STD = (0.7833, 0.6712, 0.6026), (0.2139, 0.2472, 0.2571)  # run.sh:5


def _src(B, Hs, Ws, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(B, Hs // 8 + 1, Ws // 8 + 1, 3, generator=g)
    img = torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=(Hs, Ws), mode="bilinear", align_corners=False)
    img = (img + 0.15 * torch.rand(B, 3, Hs, Ws, generator=g)).clamp(0, 1)
    return (img.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous()


def test_parameter_sampling_follows_torchvision_rules():
    from sm3hip.augment import SimCLRAugment
    aug = SimCLRAugment((224, 224), MEAN, STD)
    g = torch.Generator().manual_seed(0)
    B, Hs, Ws = 400, 462, 718  # derm7pt images after the 25 px border crop (datasets.py:520-527) are about this size
    p = aug.sample(B, Hs, Ws, g)
    i, j, h, w = p.box.unbind(1)
    assert bool(((i >= 0) & (j >= 0) & (i + h <= Hs) & (j + w <= Ws) & (h > 0) & (w > 0)).all())
    area = (h * w).double() / (Hs * Ws)
    # scale=(0.5, 1.0); on a 1.55:1 source the large crops with aspect <= 4/3 do not fit and are re-drawn (torchvision's
    # rejection loop), so the accepted areas lean towards the lower end
    assert float(area.min()) > 0.49 and float(area.max()) <= 1.0 and 0.55 < float(area.mean()) < 0.8
    ar = w.double() / h.double()
    assert float(ar.min()) > 0.74 and float(ar.max()) < 1.34                                       # ratio=(3/4, 4/3)
    applied = (p.ops != 0).any(0)
    assert 0.7 < float(applied.float().mean()) < 0.9                                               # p = 0.8
    for b in range(B):
        if bool(applied[b]):
            assert sorted(int(v) for v in p.ops[:, b]) == [1, 2, 3, 4]                             # a permutation of the four ops
    br = p.factors[p.ops == 1]
    hue = p.factors[p.ops == 4]
    assert 0.2 <= float(br.min()) and float(br.max()) <= 1.8 and -0.2 <= float(hue.min()) and float(hue.max()) <= 0.2
    assert 0.1 < float(p.gray.float().mean()) < 0.3 and 0.4 < float(p.flip.float().mean()) < 0.6
    blurred = p.sigma > 0
    assert 0.4 < float(blurred.float().mean()) < 0.6 and float(p.sigma[blurred].min()) >= 0.1 and float(p.sigma.max()) <= 2.0
    # an image no crop attempt fits (extreme aspect ratio) takes the central-crop fallback
    from sm3hip.augment import resized_crop_params
    box = resized_crop_params(4, 40, 400, (0.5, 1.0), (3 / 4, 4 / 3), g)
    assert bool((box[:, 2] == 40).all()) and bool((box[:, 3] == int(round(40 * 4 / 3))).all())


def test_oracle_identities():
    src = _src(1, 48, 64, 1)[0]
    img = A.resized_crop(src, (0, 0, 48, 64), False, 48, 64)
    assert torch.allclose(img, src.double().permute(2, 0, 1) / 255.0, atol=1e-12)  # same size: the resample is the identity
    assert torch.allclose(A.color_op(img, 1, 1.0), img) and torch.allclose(A.color_op(img, 2, 1.0), img)
    assert torch.allclose(A.color_op(img, 3, 1.0), img) and torch.allclose(A.color_op(img, 4, 0.0), img, atol=1e-9)
    g = A.color_op(img, 3, 0.0)
    assert torch.allclose(g[0], g[1]) and torch.allclose(g[1], g[2])               # saturation 0 = grayscale
    assert torch.allclose(A.blur3(img, 1e-3), img, atol=1e-9)
    flat = torch.full((3, 8, 8), 0.4, dtype=torch.float64)
    assert torch.allclose(A.blur3(flat, 1.3), flat)                                # the kernel sums to 1 (reflect padding)
    half = A.resized_crop(src, (0, 0, 48, 64), False, 24, 32)                      # 2x down: antialias = 2x2 box-ish average
    assert abs(float(half.mean()) - float(img.mean())) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 64), (56, 40)])
def test_gpu_augmentation_matches_the_oracle(size):
    from sm3hip.augment import SimCLRAugment
    B, Hs, Ws = 12, 150, 210
    src = _src(B, Hs, Ws, 7)
    aug = SimCLRAugment(size, MEAN, STD)
    g = torch.Generator().manual_seed(11)
    for trial in range(3):
        p = aug.sample(B, Hs, Ws, g)
        if trial == 0:  # make sure every op, both flags and both blur branches occur in the checked batch
            p.ops[:, 0] = torch.tensor([4, 2, 1, 3], dtype=torch.int32)
            p.factors[:, 0] = torch.tensor([0.17, 1.6, 0.3, 0.05])
            p.ops[:, 1] = 0
            p.gray[2], p.flip[3], p.sigma[0], p.sigma[1] = 1, 1, 0.0, 1.7
            p.box[4] = torch.tensor([0, 0, Hs, Ws], dtype=torch.int32)                  # full image: 2.3x / 3.3x downscale
            p.box[5] = torch.tensor([10, 20, size[0] // 2, size[1] // 2], dtype=torch.int32)  # upscale
        out = aug.apply(src.cuda(), p).cpu().double()
        assert tuple(out.shape) == (B, 3) + tuple(size)
        for b in range(B):
            ref = A.augment_one(src[b], p.box[b], bool(p.flip[b]), p.ops[:, b], p.factors[:, b], bool(p.gray[b]),
                                float(p.sigma[b]), MEAN, STD, *size)
            err = (out[b] - ref).abs()
            assert float(err.mean()) < 1e-5 and float(err.max()) < 2e-3, (trial, b, float(err.mean()), float(err.max()))


@pytest.mark.gpu
def test_two_views_feed_the_encoder():
    """NViewsTransform(aug, 2) on the GPU -> the engine's stem (what tools/backbone_train.py --data-name synthetic-u8 does)."""
    from sm3hip.augment import SimCLRAugment
    from sm3hip.trainer import SM3Trainer
    from src.models.simclr import SimCLRSkinV32
    torch.manual_seed(0)
    aug = SimCLRAugment((64, 64), MEAN, STD)
    g = torch.Generator().manual_seed(3)
    derm_src, clinic_src = _src(8, 100, 140, 1).cuda(), _src(8, 100, 140, 2).cuda()
    derm, clinic = aug(derm_src, g), aug(clinic_src, g)
    assert len(derm) == 2 and derm[0].shape == (8, 3, 64, 64) and not torch.equal(derm[0], derm[1])
    model = SimCLRSkinV32("resnet50", None, 128, 0.1)
    model.sm3_dtype = torch.bfloat16
    model.cuda()
    loss = SM3Trainer(model, lr=1e-4).step(derm, clinic)
    torch.cuda.synchronize()
    assert math.isfinite(float(loss))
