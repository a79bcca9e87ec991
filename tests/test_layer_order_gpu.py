"""The K order of a convolution is a property of the LAYER, never of the grid.

A stride-1 3x3 launch can reach four kernels -- the halo-resident one, the general one-stage gather, the 4-stage "deep"
kernel of small grids (at most 256 workgroups) and 64- instead of 128-column tiles (at most 96 tiles) -- and which one
runs depends on the batch.  The halo kernel sums chunk outer / tap inner; since round 5 the others do the same for the
launches the halo kernel could take (ConvParams::kord), so the same image gives the same output bits at batch N, 2N, 32N,
alone or beside other images: the precondition of "both views in one batch == per-view passes" (src/models/simclr.py:54-60
feeds each view through the encoder on its own) at ANY batch size.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _run(ops, code, dt, N, H, W, Ci, Co, x_all, w, dy_all, wdg, add_all, msk_all):
    D = torch.device(DEV)
    M = N * H * W
    d = ops.fwd_desc(code, N, H, W, Ci, Co, 3, 1, 1)
    x = x_all[:M].contiguous()
    y = torch.empty(M, Co, dtype=dt, device=D)
    part = torch.zeros(ops.conv_partial_rows(d) * 2 * Co, device=D)
    ops.conv_gemm(d, x, w, y, None, part)
    descs, _ = ops.dgrad_descs(code, N, H, W, Ci, Co, 3, 1, 1)
    dz = torch.empty(M, Ci, dtype=dt, device=D)
    fp = torch.zeros(sum(ops.conv_partial_rows(dd) for dd in descs) * 2 * Ci, device=D)
    off = 0
    for dd in descs:
        off += ops.conv_dgrad_bnfuse(dd, dy_all[:M].contiguous(), wdg, dz, add_all[:M].contiguous(),
                                     msk_all[:M * Ci // 8].contiguous(), None, None, None, fp, off)
    torch.cuda.synchronize()
    return y, part.view(-1, 2, Co), dz, fp.view(-1, 2, Ci)


# (H, W, Ci, Co, batches): the batches straddle the deep threshold (256 workgroups), the narrow-tile threshold (96 tiles of
# 128 columns) and, for W = 56 with 128 output channels (layer 2 at 448 x 448), the LDS fit of the halo image
CASES = [
    (14, 14, 256, 256, (2, 64, 128, 200)),   # layer 3: 16 workgroups (narrow + deep) / 196 (deep) / 392, 614 (halo)
    (7, 7, 512, 512, (4, 128, 300)),          # layer 4: narrow + deep / deep / halo
    (28, 28, 128, 128, (1, 12, 40, 48)),      # layer 2: 7 tiles / 74 (narrow) / 245 (deep) / 294 (halo)
    (56, 56, 128, 128, (1, 3, 8, 16)),        # 448 x 448 layer 2: the image only fits with 64-column tiles
    (56, 56, 64, 64, (1, 8, 16)),             # layer 1: 64-column tiles always, one chunk
    (112, 112, 64, 64, (1, 4)),               # 448 x 448 layer 1: never fits, general gather at every size
]


@pytest.mark.parametrize("deep", ["1", "0"], ids=["deep_on", "deep_off"])
@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_same_image_same_bits_at_every_batch_size(dtname, deep, monkeypatch):
    from sm3hip import ops
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtname]
    code = ops.dtype_code(dt)
    monkeypatch.setenv("SM3_CONV_DEEP", deep)
    D = torch.device(DEV)
    for ci_, (H, W, Ci, Co, batches) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + ci_)
        Nmax = max(batches)
        Mmax = Nmax * H * W
        x_all = torch.randn(Mmax, Ci, generator=g).to(dt).to(D)
        dy_all = torch.randn(Mmax, Co, generator=g).to(dt).to(D)
        add_all = torch.randn(Mmax, Ci, generator=g).to(dt).to(D)
        msk_all = torch.randint(0, 256, (Mmax * Ci // 8,), generator=g, dtype=torch.uint8).to(D)
        w = (torch.randn(Co, 9 * Ci, generator=g) / math.sqrt(9 * Ci)).to(dt).to(D)
        wdg = (torch.randn(Ci, 9 * Co, generator=g) / math.sqrt(9 * Co)).to(dt).to(D)
        outs = {N: _run(ops, code, dt, N, H, W, Ci, Co, x_all, w, dy_all, wdg, add_all, msk_all) for N in batches}
        N0 = batches[0]
        M0 = N0 * H * W
        full = M0 // 128  # partial-sum rows (one per 128-row tile) that lie wholly inside the shared images
        for N in batches[1:]:
            what = (dtname, (H, W, Ci, Co), N0, N)
            assert torch.equal(outs[N][0][:M0], outs[N0][0]), ("y",) + what
            assert torch.equal(outs[N][2][:M0], outs[N0][2]), ("dz",) + what
            if full:
                assert torch.equal(outs[N][1][:full], outs[N0][1][:full]), ("bn partial sums",) + what
                assert torch.equal(outs[N][3][:full], outs[N0][3][:full]), ("fused phase-1 partial sums",) + what
        # ... and the values are the convolution's: the smallest batch against fp64
        xs = x_all[:M0].cpu().double().view(N0, H, W, Ci).permute(0, 3, 1, 2)
        ws = w.cpu().double().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xs, ws, padding=1).permute(0, 2, 3, 1).reshape(M0, Co)
        ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        assert float((outs[N0][0].cpu().double() - ref).abs().max()) < 4 * ulp * float(ref.abs().max())
