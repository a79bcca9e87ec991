"""Pins oracle/augment_oracle.py -- the float restatement the GPU augmentation kernels are tested against -- to PIL
itself, which is what the reference's transform chain (tools/backbone_train.py:448-466: torchvision 0.13 transforms applied
to PIL images from src/utils/data/functional.py:38-39) runs on for five of its six image operations:

    RandomResizedCrop   Image.crop + Image.resize(BILINEAR)           (torchvision F_pil.crop / resize)
    brightness          ImageEnhance.Brightness                       (F_pil.adjust_brightness)
    contrast            ImageEnhance.Contrast                         (F_pil.adjust_contrast)
    saturation          ImageEnhance.Color                            (F_pil.adjust_saturation)
    RandomGrayscale     Image.convert("L")                            (F_pil.to_grayscale)
    hue                 convert("HSV"), uint8 shift of H, convert("RGB")   (F_pil.adjust_hue)

PIL keeps uint8 between operations; the oracle is the un-quantised arithmetic of the same operations, so each operation
must agree with PIL's result on a uint8 input to 1 LSB of rounding plus what PIL itself truncates (bounds stated per test;
hue: to PIL's own 8-bit quantisation of H, S and V).  NOT pinned: the 3x3 GaussianBlur, which torchvision 0.13 computes with its TENSOR kernel even for PIL inputs
(F_t.gaussian_blur on pil_to_tensor(img)) -- torchvision is not in this image; the oracle restates it from its published
algorithm.  CPU only."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

from oracle import augment_oracle as A


def _img(seed, h=97, w=113):
    g = np.random.default_rng(seed)
    # smooth structure + noise: realistic local contrast, every channel value range used
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0 - c) for c in range(3)], -1)
    return np.clip(base + g.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


def _t01(u8):  # [H, W, 3] uint8 -> [3, H, W] float64 in [0, 1]
    return torch.from_numpy(u8.astype(np.float64) / 255.0).permute(2, 0, 1)


def _lsb(t01, pil_img):
    """max |oracle * 255 - PIL| in LSB, oracle NOT rounded: 0.5 is pure quantisation."""
    want = np.asarray(pil_img).astype(np.float64)
    got = (t01.clamp(0, 1) * 255.0).permute(1, 2, 0).numpy() if t01.dim() == 3 else (t01.clamp(0, 1) * 255.0).numpy()
    return float(np.abs(got - want).max()), float(np.abs(got - want).mean())


@pytest.mark.parametrize("box,size", [((5, 9, 80, 96), (64, 64)),       # down-scaling: the antialiased (support > 1) case
                                      ((20, 30, 40, 33), (64, 64)),      # up-scaling
                                      ((0, 0, 97, 113), (48, 80)),       # whole image, anisotropic
                                      ((13, 7, 64, 64), (64, 64))])      # identity resample
def test_resized_crop_against_pil(box, size):
    src = _img(1)
    i, j, h, w = box
    H, W = size
    pil = Image.fromarray(src).crop((j, i, j + w, i + h)).resize((W, H), Image.BILINEAR)
    mx, mean = _lsb(A.resized_crop(torch.from_numpy(src), box, False, H, W), pil)
    assert mx <= 1.0 and mean < 0.4, (mx, mean)   # PIL: 8-bit fixed-point coefficients + rounding after each axis
    flipped = A.resized_crop(torch.from_numpy(src), box, True, H, W)
    assert _lsb(flipped, pil.transpose(Image.FLIP_LEFT_RIGHT))[0] <= 1.0


@pytest.mark.parametrize("f", [0.2, 0.7, 1.0, 1.4, 1.8])                  # ColorJitter(0.8, ...): factors in [0.2, 1.8]
def test_brightness_contrast_saturation_against_pil(f):
    src = _img(2)
    pil, t = Image.fromarray(src), _t01(src)
    for op, enh in ((1, ImageEnhance.Brightness), (2, ImageEnhance.Contrast), (3, ImageEnhance.Color)):
        mx, mean = _lsb(A.color_op(t, op, f), enh(pil).enhance(f))
        # PIL's Image.blend TRUNCATES the blended value to uint8 (up to 1 LSB below the exact one, 0.5 on average) and
        # blends against a uint8-rounded degenerate image (integer grey mean / uint8 luma, ITU-R 601 in 16-bit fixed point
        # where torchvision's tensor path has 0.2989 / 0.587 / 0.114): another 0.5 |1 - f| at most
        assert mx <= 1.0 + 0.5 * abs(1.0 - f) and mean < 0.75, (op, f, mx, mean)


def test_grayscale_against_pil():
    src = _img(3)
    mx, mean = _lsb(A.gray(_t01(src)), Image.fromarray(src).convert("L"))
    assert mx <= 1.0 and mean < 0.3, (mx, mean)


@pytest.mark.parametrize("f", [-0.2, -0.05, 0.0, 0.1, 0.2])              # ColorJitter(..., hue=0.2)
def test_hue_against_pil(f):
    src = _img(4)
    h, s, v = Image.fromarray(src).convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        nh += np.uint8(int(f * 255) & 0xFF)                              # torchvision F_pil.adjust_hue: uint8 wrap-around
    pil = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    mx, mean = _lsb(A.color_op(_t01(src), 4, f), pil)
    # PIL stores H, S and V as uint8: one step of H is 6/256 of a colour-wheel sector, i.e. up to ~6 LSB of the fastest
    # channel at full saturation and value, plus the truncation of int(f * 255).  The mean shows the agreement.
    assert mx <= 12.0 and mean < 1.5, (f, mx, mean)
