"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, fp64) of the reference's augmentation chain
(tools/backbone_train.py:448-466) for given random parameters.

The arithmetic lives in un-vendored dependencies of the reference: torchvision (pinned torchvision==0.13.0,
requirements.txt:3; absent from this image) and PIL.  Restated here from their published algorithms:
  * resized crop: PIL ImagingResample bilinear = triangle filter with support max(scale, 1) per axis, weights
    normalised per output pixel (the antialiasing torchvision's tensor path calls antialias=True);
  * ColorJitter ops: torchvision/transforms/functional_tensor.py  _blend, rgb_to_grayscale (0.2989, 0.587, 0.114),
    adjust_brightness / contrast / saturation / hue with _rgb2hsv / _hsv2rgb;
  * GaussianBlur: _get_gaussian_kernel1d over linspace(-1, 1, 3), reflect padding;
  * ToTensor (/255) and Normalize.
PINNED TO PIL for five of the six image operations (tests/test_augment_pil.py, CPU): the reference feeds PIL images, on
which torchvision 0.13's RandomResizedCrop, brightness / contrast / saturation / hue and RandomGrayscale ARE PIL calls
(Image.crop + resize(BILINEAR), ImageEnhance.Brightness / Contrast / Color, the HSV round trip, convert("L")); resized_crop,
color_op and gray here agree with PIL 12.2 on uint8 inputs to 1 LSB of rounding plus PIL's own truncations (hue: PIL's
8-bit H, S, V).  This float chain is the un-quantised form of that pipeline: PIL rounds to uint8 after every operation, the
chain only at the end.  UNPINNED: blur3 -- torchvision's GaussianBlur uses its tensor kernel even on PIL inputs, and
torchvision cannot be imported here; restated from its published algorithm.
"""
import torch


def _tri_weights(out_size, in_size):
    scale = in_size / out_size
    fs = max(scale, 1.0)
    rows = []
    for o in range(out_size):
        c = (o + 0.5) * scale
        lo, hi = max(0, int(c - fs + 0.5)), min(in_size, int(c + fs + 0.5))
        w = torch.zeros(in_size, dtype=torch.float64)
        for k in range(lo, hi):
            w[k] = max(0.0, 1.0 - abs((k - c + 0.5) / fs))
        rows.append(w / w.sum())
    return torch.stack(rows)  # [out, in]


def resized_crop(src_u8, box, flip, H, W):
    """src_u8 [Hs, Ws, 3] uint8 -> [3, H, W] fp64 in [0, 1]."""
    i, j, h, w = [int(v) for v in box]
    crop = src_u8[i:i + h, j:j + w].double().permute(2, 0, 1)  # [3, h, w]
    wy, wx = _tri_weights(H, h), _tri_weights(W, w)
    out = torch.einsum("yh,chw,xw->cyx", wy, crop, wx) / 255.0
    return out.flip(-1) if flip else out


def gray(img):
    return 0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]


def _blend(a, b, f):
    return (f * a + (1.0 - f) * b).clamp(0, 1)


def _rgb2hsv(img):
    r, g, b = img
    maxc, minc = img.max(0).values, img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return h, s, maxc


def _hsv2rgb(h, s, v):
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int64) % 6
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    sel = lambda opts: sum((i == k) * opts[k] for k in range(6))
    return torch.stack([sel((v, q, p, p, t, v)), sel((t, v, v, q, p, p)), sel((p, p, t, v, v, q))])


def color_op(img, op, f):
    if op == 0:
        return img
    if op == 1:
        return _blend(img, torch.zeros_like(img), f)
    if op == 2:
        return _blend(img, gray(img).mean(), f)
    if op == 3:
        return _blend(img, gray(img).unsqueeze(0), f)
    h, s, v = _rgb2hsv(img)
    return _hsv2rgb((h + f) % 1.0, s, v)


def blur3(img, sigma):
    x = torch.linspace(-1.0, 1.0, 3, dtype=torch.float64)
    k = torch.exp(-0.5 * (x / sigma) ** 2)
    k = k / k.sum()
    pad = torch.nn.functional.pad(img.unsqueeze(0), (1, 1, 1, 1), mode="reflect")
    k2 = (k[:, None] * k[None, :]).expand(3, 1, 3, 3)
    return torch.nn.functional.conv2d(pad, k2, groups=3)[0]


def augment_one(src_u8, box, flip, ops, factors, to_gray, sigma, mean, std, H, W):
    img = resized_crop(src_u8, box, flip, H, W)
    for op, f in zip(ops, factors):
        img = color_op(img, int(op), float(f))
    if to_gray:
        img = gray(img).unsqueeze(0).expand(3, -1, -1)
    if sigma > 0:
        img = blur3(img.contiguous(), float(sigma))
    m = torch.tensor(mean, dtype=torch.float64).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float64).view(3, 1, 1)
    return (img - m) / s
