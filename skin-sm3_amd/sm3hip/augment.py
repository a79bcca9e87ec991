"""GPU-side SimCLR augmentation of the SM3 pre-training input: the transform chain of
tools/backbone_train.py:448-466

    RandomResizedCrop(img_sz, scale=(0.5, 1.0)) -> RandomApply([ColorJitter(0.8, 0.8, 0.8, 0.2)], p=0.8)
    -> RandomGrayscale(p=0.2) -> RandomHorizontalFlip() -> RandomApply([GaussianBlur((3, 3), (0.1, 2.0))], p=0.5)
    -> ToTensor() -> Normalize(mean, std)                       wrapped in NViewsTransform(., 2) (functional.py:43-49)

on a batch of decoded RGB images already resident in HBM ([B, Hs, Ws, 3] uint8), through csrc/augment.hip.  This module
is the host half: it draws the random parameters with torchvision 0.13's rules (RandomResizedCrop.get_params: 10 tries
of area ~ U(scale) x aspect ~ logU(3/4, 4/3), central-crop fallback; ColorJitter.get_params: a random permutation of
the four ops, factors U(1-s, 1+s) / hue U(-h, h); GaussianBlur.get_params: sigma ~ U(0.1, 2.0)) and sequences the
kernels.  The flip is folded into the resample (it commutes with every later op: they are per-pixel, a global mean, or a
symmetric filter with reflect padding).  Arithmetic is float (torchvision's tensor code path); the reference's PIL path
rounds to uint8 after every op."""
import ctypes as C
import math

import torch

from . import _lib, ops

OPS = {"brightness": 1, "contrast": 2, "saturation": 3, "hue": 4}


class AugParams:
    """Per-sample random parameters of one view batch (host tensors)."""
    __slots__ = ("box", "flip", "ops", "factors", "gray", "sigma")


def resized_crop_params(B, Hs, Ws, scale, ratio, gen):
    """torchvision.transforms.RandomResizedCrop.get_params for every sample of the batch -> int32 [B, 4] (i, j, h, w)."""
    box = torch.empty(B, 4, dtype=torch.int32)
    area = Hs * Ws
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for b in range(B):
        done = False
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(scale[0], scale[1], generator=gen))
            ar = math.exp(float(torch.empty(1).uniform_(lo, hi, generator=gen)))
            w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < w <= Ws and 0 < h <= Hs:
                i = int(torch.randint(0, Hs - h + 1, (1,), generator=gen))
                j = int(torch.randint(0, Ws - w + 1, (1,), generator=gen))
                box[b] = torch.tensor([i, j, h, w])
                done = True
                break
        if not done:  # central crop
            in_ratio = Ws / Hs
            if in_ratio < min(ratio):
                w, h = Ws, int(round(Ws / min(ratio)))
            elif in_ratio > max(ratio):
                h, w = Hs, int(round(Hs * max(ratio)))
            else:
                w, h = Ws, Hs
            box[b] = torch.tensor([(Hs - h) // 2, (Ws - w) // 2, h, w])
    return box


class SimCLRAugment:
    def __init__(self, size, mean, std, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), jitter=(0.8, 0.8, 0.8, 0.2),
                 p_jitter=0.8, p_gray=0.2, p_flip=0.5, p_blur=0.5, sigma=(0.1, 2.0)):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.scale, self.ratio, self.jitter = scale, ratio, jitter
        self.p_jitter, self.p_gray, self.p_flip, self.p_blur, self.sigma = p_jitter, p_gray, p_flip, p_blur, sigma

    def sample(self, B, Hs, Ws, gen=None):
        """Draw one view's parameters (CPU generator: the reference samples on DataLoader workers)."""
        u = lambda n=B: torch.rand(n, generator=gen)
        p = AugParams()
        p.box = resized_crop_params(B, Hs, Ws, self.scale, self.ratio, gen)
        apply_j = u() < self.p_jitter
        p.ops = torch.zeros(4, B, dtype=torch.int32)      # [position][sample]
        p.factors = torch.ones(4, B, dtype=torch.float32)
        bj, cj, sj, hj = self.jitter
        for b in range(B):
            order = torch.randperm(4, generator=gen)        # ColorJitter.get_params
            f = [float(torch.empty(1).uniform_(max(0.0, 1 - bj), 1 + bj, generator=gen)),
                 float(torch.empty(1).uniform_(max(0.0, 1 - cj), 1 + cj, generator=gen)),
                 float(torch.empty(1).uniform_(max(0.0, 1 - sj), 1 + sj, generator=gen)),
                 float(torch.empty(1).uniform_(-hj, hj, generator=gen))]
            if bool(apply_j[b]):
                for pos in range(4):
                    fn = int(order[pos])                    # 0 brightness, 1 contrast, 2 saturation, 3 hue
                    p.ops[pos, b] = fn + 1
                    p.factors[pos, b] = f[fn]
        p.gray = (u() < self.p_gray).to(torch.uint8)
        p.flip = (u() < self.p_flip).to(torch.uint8)
        blur = u() < self.p_blur
        sig = torch.empty(B).uniform_(self.sigma[0], self.sigma[1], generator=gen)
        p.sigma = torch.where(blur, sig, torch.zeros(B))    # 0 = no blur
        return p

    def apply(self, src, params):
        """src: [B, Hs, Ws, 3] uint8 on the GPU -> [B, 3, H, W] fp32 normalised (what the encoder's stem reads)."""
        if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3 or not src.is_contiguous():
            raise ValueError("augmentation input must be a contiguous [B, Hs, Ws, 3] uint8 CUDA tensor")
        B, Hs, Ws, _ = src.shape
        H, W = self.size
        dev = src.device
        box = params.box.to(dev)
        if params.box.shape != (B, 4) or bool((params.box[:, 2] <= 0).any()) or bool((params.box[:, 3] <= 0).any()) or \
                bool((params.box[:, 0] < 0).any()) or bool((params.box[:, 1] < 0).any()) or \
                bool((params.box[:, 0] + params.box[:, 2] > Hs).any()) or bool((params.box[:, 1] + params.box[:, 3] > Ws).any()):
            raise ValueError("crop box outside the source image")
        lib, st = _lib.load(), ops._stream()
        # device copies of the parameters stay referenced until every launch is enqueued (a temporary handed to ctypes
        # would be recycled by the caching allocator for the next temporary)
        flip, gray, sigma = params.flip.to(dev), params.gray.to(dev), params.sigma.to(dev)
        opsd, facd = params.ops.contiguous().to(dev), params.factors.contiguous().to(dev)
        img = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        _lib.check(lib.sm3_aug_resized_crop(ops._ptr(src), B, Hs, Ws, ops._ptr(box), ops._ptr(flip), ops._ptr(img), H, W, st),
                   "sm3_aug_resized_crop")
        gm = torch.empty(B, dtype=torch.float32, device=dev)
        for pos in range(4):
            if bool((params.ops[pos] != 0).any()):
                _lib.check(lib.sm3_aug_color_op(ops._ptr(img), B, H, W, ops._ptr(opsd[pos]), ops._ptr(facd[pos]), ops._ptr(gm),
                                                st), "sm3_aug_color_op")
        out = torch.empty_like(img)
        m3, s3 = (C.c_float * 3)(*self.mean), (C.c_float * 3)(*self.std)
        _lib.check(lib.sm3_aug_finish(ops._ptr(img), B, H, W, ops._ptr(gray), ops._ptr(sigma), m3, s3, ops._ptr(out), st),
                   "sm3_aug_finish")
        return out

    def __call__(self, src, gen=None, n_views=2):
        """NViewsTransform(self, n_views): independent parameters per view (functional.py:43-49)."""
        B, Hs, Ws, _ = src.shape
        return [self.apply(src, self.sample(B, Hs, Ws, gen)) for _ in range(n_views)]
