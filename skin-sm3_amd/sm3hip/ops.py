"""Tensor-level wrappers over the C ABI (include/sm3_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every kernel is ours.  Each wrapper
checks on the host that shapes/dtypes/devices match what the kernel's grid assumes before launching.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, SM3_BF16, SM3_F16, SM3_F32, check

TORCH_DTYPE = {SM3_F32: torch.float32, SM3_BF16: torch.bfloat16, SM3_F16: torch.float16}
K_CHUNK = {SM3_F32: 32, SM3_BF16: 64, SM3_F16: 64}  # elements per 128-byte K chunk


_PROFILER = None


def set_profiler(p):
    """Install a sm3hip.profiler.Profiler (or None): wrappers then bracket their launches with HIP events."""
    global _PROFILER
    _PROFILER = p


class _prof:
    """with _prof(tag, flops, bytes): launch  -- no-op unless a profiler that wants `tag` is installed."""
    __slots__ = ("tag", "flops", "nbytes", "start")

    def __init__(self, tag, flops=0.0, nbytes=0.0):
        self.tag, self.flops, self.nbytes, self.start = tag, flops, nbytes, None

    def __enter__(self):
        p = _PROFILER
        if p is not None and p.wants(self.tag):
            self.start = p.begin()

    def __exit__(self, *exc):
        if self.start is not None:
            _PROFILER.end(self.tag, self.flops, self.nbytes, self.start)
        return False


def _sz(dtype):
    return 4 if dtype == SM3_F32 else 2


# torch.cuda.current_stream() costs ~8 us of host time and every wrapper needs the handle: 2 400 calls = 30 % of the
# host side of a step.  A caller that owns the stream context (the engine: one scope per lane entry / step) pins the
# raw handle with stream_scope(); outside such a scope the handle is looked up per launch as before.
_PINNED_STREAM = None


class stream_scope:
    """with ops.stream_scope(): every launch inside goes to torch's current stream AS OF ENTRY (raw handle cached).
    Nest a new scope whenever the current stream changes (torch.cuda.stream(...))."""

    def __enter__(self):
        global _PINNED_STREAM
        self.prev = _PINNED_STREAM
        _PINNED_STREAM = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        return False


def _stream_handle():
    if _PINNED_STREAM is not None:
        return _PINNED_STREAM
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return C.c_void_p(_stream_handle())


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (the SM3 HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")


def dtype_code(torch_dtype):
    if torch_dtype == torch.float32:
        return SM3_F32
    if torch_dtype == torch.bfloat16:
        return SM3_BF16
    if torch_dtype == torch.float16:
        return SM3_F16
    raise ValueError(f"unsupported activation dtype {torch_dtype}")


# ------------------------------------------------------------------------------------------
# conv descriptors
# ------------------------------------------------------------------------------------------
def make_desc(dtype, N, Hi, Wi, Ci, Ho, Wo, Co, sy, sx, taps, w_row_stride, Hout=None, Wout=None,
              osy=1, osx=1, ooy=0, oox=0):
    """taps: list of (dy, dx, wtap)."""
    d = ConvDesc()
    d.dtype = dtype
    d.N, d.Hi, d.Wi, d.Ci = N, Hi, Wi, Ci
    d.Ho, d.Wo, d.Co = Ho, Wo, Co
    d.sy, d.sx = sy, sx
    d.ntaps = len(taps)
    if not 1 <= len(taps) <= _lib.MAX_TAPS:
        raise ValueError("1..9 taps")
    for i, (dy, dx, wt) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = dy, dx, wt
    d.w_row_stride = w_row_stride
    d.Hout = Ho if Hout is None else Hout
    d.Wout = Wo if Wout is None else Wout
    d.osy, d.osx, d.ooy, d.oox = osy, osx, ooy, oox
    return d


def fwd_desc(dtype, N, Hi, Wi, Ci, Co, k, stride, pad):
    """Forward conv k x k / stride / pad over NHWC x with OHWI weights [Co][k*k][Ci]."""
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    taps = [(kh - pad, kw - pad, kh * k + kw) for kh in range(k) for kw in range(k)]
    return make_desc(dtype, N, Hi, Wi, Ci, Ho, Wo, Co, stride, stride, taps, k * k * Ci)


def dgrad_descs(dtype, N, Hi, Wi, Ci, Co, k, stride, pad):
    """Data gradient of fwd_desc(...) as gather-GEMMs over dY [N,Ho,Wo,Co] with the transposed filter
    bank w_dgrad [Ci][k*k][Co]: one descriptor per output-parity class (stride^2 of them), each with
    only the taps that hit real dY pixels.  Returns (descs, covers_everything)."""
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    descs, full = [], True
    for py in range(stride):
        for px in range(stride):
            Hq = (Hi - py + stride - 1) // stride
            Wq = (Wi - px + stride - 1) // stride
            if Hq <= 0 or Wq <= 0:
                continue
            taps = [((py + pad - kh) // stride, (px + pad - kw) // stride, kh * k + kw)
                    for kh in range(k) for kw in range(k)
                    if (py + pad - kh) % stride == 0 and (px + pad - kw) % stride == 0]
            if not taps:
                full = False
                continue
            descs.append(make_desc(dtype, N, Ho, Wo, Co, Hq, Wq, Ci, 1, 1, taps, k * k * Co,
                                   Hout=Hi, Wout=Wi, osy=stride, osx=stride, ooy=py, oox=px))
    return descs, full


def conv_partial_rows(desc):
    return _lib.load().sm3_conv_partial_rows(C.byref(desc))


def _conv_tag(desc):
    """Profiler class = the tile the library picks (csrc/conv_igemm.hip, conv_gather_gemm_impl): 128 x 64 for Co <= 64 and
    for the small-M Linears whose 128-column grid would leave most CUs idle (those also take the 4-stage K-loop)."""
    M = desc.N * desc.Ho * desc.Wo
    nsteps = desc.ntaps * (desc.Ci * _sz(desc.dtype) // 128)
    tiles128 = ((M + 127) // 128) * ((desc.Co + 127) // 128)
    return "conv_gemm_128x64" if desc.Co <= 64 or (tiles128 <= 96 and nsteps >= 6) else "conv_gemm_128x128"


def conv_gemm(desc, x, w, y, addend=None, partials=None):
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(w, tdt, "w"); _chk(y, tdt, "y"); _chk(addend, tdt, "addend")
    _chk(partials, torch.float32, "partials")
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError(f"x has {x.numel()} elements, descriptor says {desc.N}x{desc.Hi}x{desc.Wi}x{desc.Ci}")
    if y.numel() != desc.N * desc.Hout * desc.Wout * desc.Co:
        raise ValueError("y size does not match descriptor")
    if addend is not None and addend.numel() != y.numel():
        raise ValueError("addend size does not match y")
    need_w = (desc.Co - 1) * desc.w_row_stride + max(desc.wtap[i] for i in range(desc.ntaps)) * desc.Ci + desc.Ci
    if w.numel() < need_w:
        raise ValueError("w too small for descriptor")
    if desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError(f"Ci={desc.Ci} is not a multiple of {K_CHUNK[desc.dtype]}")
    if partials is not None and partials.numel() < conv_partial_rows(desc) * 2 * desc.Co:
        raise ValueError("partials workspace too small")
    M = desc.N * desc.Ho * desc.Wo
    flops = 2.0 * M * desc.Co * desc.ntaps * desc.Ci
    sz = _sz(desc.dtype)
    nbytes = sz * (min(x.numel(), M * desc.ntaps * desc.Ci) + M * desc.Co * (2 if addend is not None else 1)
                   + desc.Co * desc.ntaps * desc.Ci)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}_s{desc.osy}"
    with _prof(tag, flops, nbytes):
        check(_lib.load().sm3_conv_gather_gemm(C.byref(desc), _ptr(x), _ptr(w), _ptr(y), _ptr(addend),
                                               _ptr(partials), _stream()), "sm3_conv_gather_gemm")


def conv_dgrad_bnfuse(desc, dy_in, w_dgrad, dz_out, addend, mask, bn_x, mean, invstd, partials, row_offset,
                      views=1, row_offset_view1=0, addend_sparse=None):
    """Data-gradient launch that also masks with the producer BN's ReLU bits and emits its backward partial sums
    (see sm3_conv_dgrad_bnfuse).  Returns the number of partial rows this launch wrote (both views together).
    views=2: the launch's rows are two views back to back (each a multiple of 128 rows), mean/invstd are [2][C],
    view 0's tiles write partial rows from row_offset, view 1's from row_offset_view1."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(dy_in, tdt, "dy_in"); _chk(w_dgrad, tdt, "w"); _chk(dz_out, tdt, "dz_out"); _chk(addend, tdt, "addend")
    _chk(bn_x, tdt, "bn_x"); _chk(mask, torch.uint8, "mask"); _chk(mean, torch.float32); _chk(invstd, torch.float32)
    _chk(partials, torch.float32, "partials")
    if dy_in.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("dy_in size does not match descriptor")
    n_out = desc.N * desc.Hout * desc.Wout * desc.Co
    if dz_out.numel() != n_out or (bn_x is not None and bn_x.numel() != n_out):
        raise ValueError("dz_out / bn_x size does not match descriptor")
    if addend_sparse is not None:  # compact [N, Hs, Ws, Co] addend for the even output positions
        hs, ws = addend_sparse
        if addend is None or (hs, ws) != ((desc.Ho + 1) // 2, (desc.Wo + 1) // 2) or \
                addend.numel() != desc.N * hs * ws * desc.Co:
            raise ValueError("sparse addend size does not match descriptor")
    elif addend is not None and addend.numel() != n_out:
        raise ValueError("addend size does not match descriptor")
    if mask is not None and mask.numel() != n_out // (16 // _sz(desc.dtype)):
        raise ValueError("mask size mismatch")
    f = _bn_fuse_struct(desc, mask, bn_x, mean, invstd, partials, row_offset, views, row_offset_view1, addend_sparse)
    prow = conv_partial_rows(desc)
    if desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError(f"Ci={desc.Ci} is not a multiple of {K_CHUNK[desc.dtype]}")
    M = desc.N * desc.Ho * desc.Wo
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}_s{desc.osy}_fz"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (min(dy_in.numel(), M * desc.ntaps * desc.Ci)
                     + M * desc.Co * (1 + (addend is not None) + (bn_x is not None)))):
        check(_lib.load().sm3_conv_dgrad_bnfuse(C.byref(desc), _ptr(dy_in), _ptr(w_dgrad), _ptr(dz_out), _ptr(addend),
                                                C.byref(f), _stream()), "sm3_conv_dgrad_bnfuse")
    return prow


def _bn_fuse_struct(desc, mask, bn_x, mean, invstd, partials, row_offset, views, row_offset_view1, addend_sparse=None):
    """sm3_bn_bwd_fuse for a data-gradient launch over `desc` (sizes checked).  bn_x None: mask + sum(dz) only."""
    n_out = desc.N * desc.Hout * desc.Wout * desc.Co
    if mask is not None and mask.numel() != n_out // (16 // _sz(desc.dtype)):
        raise ValueError("mask size mismatch")
    if bn_x is not None and (mean is None or invstd is None or mean.numel() < views * desc.Co
                             or invstd.numel() < views * desc.Co):
        raise ValueError("mean/invstd too small")
    prow = conv_partial_rows(desc)
    if views == 1:
        if partials.numel() < (row_offset + prow) * 2 * desc.Co:
            raise ValueError("partials workspace too small")
    else:
        if views != 2 or (desc.N * desc.Ho * desc.Wo) % 256 or prow % 2:
            raise ValueError("two views need a multiple of 128 rows each")
        if partials.numel() < (max(row_offset, row_offset_view1) + prow // 2) * 2 * desc.Co:
            raise ValueError("partials workspace too small")
    f = _lib.BnBwdFuse()
    f.relu_mask = mask.data_ptr() if mask is not None else None
    f.x = bn_x.data_ptr() if bn_x is not None else None
    f.mean = mean.data_ptr() if bn_x is not None else None
    f.invstd = invstd.data_ptr() if bn_x is not None else None
    f.partials = partials.data_ptr()
    f.partial_row_offset = row_offset
    f.views, f.partial_row_offset_view1 = views, row_offset_view1
    f.addend_sp_h, f.addend_sp_w = addend_sparse if addend_sparse is not None else (0, 0)
    return f


def conv_dgrad_seg_bnfuse(desc, x0, w0, x1, w1, col_bias, dz_out, mask, bn_x, mean, invstd, partials, row_offset,
                          views=1, row_offset_view1=0, w_view_stride=0, w1_view_stride=0):
    """dz_out = mask(x0 w0^T + x1 w1^T + col_bias) with the producer BatchNorm's backward phase 1 in the epilogue
    (sm3_conv_dgrad_seg_bnfuse): the data gradient of conv -> BatchNorm by linearity.  desc: the 1x1 data-gradient
    descriptor of x0 ([pixels][desc.Ci]); x1: [pixels][Ci1]; w0: [views][Co][Ci], w1: [views][Co][Ci1]; col_bias:
    fp32 [views][Co].  Returns the number of partial rows written."""
    tdt = TORCH_DTYPE[desc.dtype]
    for t, n in ((x0, "x0"), (w0, "w0"), (x1, "x1"), (w1, "w1"), (dz_out, "dz_out"), (bn_x, "bn_x")):
        _chk(t, tdt, n)
    _chk(col_bias, torch.float32, "col_bias"); _chk(mask, torch.uint8, "mask"); _chk(partials, torch.float32, "partials")
    _chk(mean, torch.float32); _chk(invstd, torch.float32)
    if desc.dtype == SM3_F32:
        raise ValueError("conv_dgrad_seg_bnfuse: 16-bit activation types only")
    M = desc.N * desc.Ho * desc.Wo
    if desc.ntaps != 1 or desc.Hout != desc.Ho or desc.Wout != desc.Wo or desc.Hi != desc.Ho or desc.Wi != desc.Wo:
        raise ValueError("conv_dgrad_seg_bnfuse: 1x1 / stride-1 descriptor expected")
    if x0.numel() != M * desc.Ci or x1.numel() % M or dz_out.numel() != M * desc.Co:
        raise ValueError("conv_dgrad_seg_bnfuse: operand size does not match descriptor")
    Ci1 = x1.numel() // M
    if Ci1 % K_CHUNK[desc.dtype] or desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError("conv_dgrad_seg_bnfuse: channel counts must be multiples of the K chunk")
    if w0.numel() < (views - 1) * w_view_stride + desc.Co * desc.w_row_stride or \
            w1.numel() < (views - 1) * w1_view_stride + desc.Co * Ci1:
        raise ValueError("conv_dgrad_seg_bnfuse: weight bank too small")
    if col_bias is not None and col_bias.numel() < views * desc.Co:
        raise ValueError("conv_dgrad_seg_bnfuse: col_bias too small")
    if bn_x is not None and bn_x.numel() != dz_out.numel():
        raise ValueError("conv_dgrad_seg_bnfuse: bn_x size mismatch")
    f = _bn_fuse_struct(desc, mask, bn_x, mean, invstd, partials, row_offset, views, row_offset_view1)
    sg = _lib.ConvSeg()
    sg.x1, sg.w1, sg.Ci1 = x1.data_ptr(), w1.data_ptr(), Ci1
    sg.w_view_stride, sg.w1_view_stride = w_view_stride, w1_view_stride
    sg.col_bias = col_bias.data_ptr() if col_bias is not None else None
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.Ci}+{Ci1}_N{desc.Co}_seg_fz"
    # algorithmic work = the convolution's data gradient (K = desc.Ci); the second segment is this implementation's cost
    with _prof(tag, 2.0 * M * desc.Co * desc.Ci,
               sz * (x0.numel() + x1.numel() + M * desc.Co * (1 + (bn_x is not None)))):
        check(_lib.load().sm3_conv_dgrad_seg_bnfuse(C.byref(desc), _ptr(x0), _ptr(w0), C.byref(sg), _ptr(dz_out), None,
                                                    C.byref(f), _stream()), "sm3_conv_dgrad_seg_bnfuse")
    return conv_partial_rows(desc)


def conv_gemm_seg(desc, x0, w0, x1, w1, col_bias, y, addend=None, views=1, w_view_stride=0, w1_view_stride=0):
    """y = x0 w0^T + x1 w1^T + col_bias (+ addend, which may be y itself) with per-view banks (sm3_conv_gather_gemm_seg):
    the data gradient of a downsample conv -> BatchNorm pair by linearity."""
    tdt = TORCH_DTYPE[desc.dtype]
    for t, n in ((x0, "x0"), (w0, "w0"), (x1, "x1"), (w1, "w1"), (y, "y"), (addend, "addend")):
        _chk(t, tdt, n)
    _chk(col_bias, torch.float32, "col_bias")
    if desc.dtype == SM3_F32:
        raise ValueError("conv_gemm_seg: 16-bit activation types only")
    M = desc.N * desc.Ho * desc.Wo
    if desc.ntaps != 1 or desc.Hout != desc.Ho or desc.Wout != desc.Wo or desc.Hi != desc.Ho or desc.Wi != desc.Wo:
        raise ValueError("conv_gemm_seg: 1x1 / stride-1 descriptor expected")
    if x0.numel() != M * desc.Ci or x1.numel() % M or y.numel() != M * desc.Co or \
            (addend is not None and addend.numel() != y.numel()):
        raise ValueError("conv_gemm_seg: operand size does not match descriptor")
    Ci1 = x1.numel() // M
    if Ci1 % K_CHUNK[desc.dtype] or desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError("conv_gemm_seg: channel counts must be multiples of the K chunk")
    if w0.numel() < (views - 1) * w_view_stride + desc.Co * desc.w_row_stride or \
            w1.numel() < (views - 1) * w1_view_stride + desc.Co * Ci1 or \
            (col_bias is not None and col_bias.numel() < views * desc.Co):
        raise ValueError("conv_gemm_seg: weight bank / col_bias too small")
    if views > 1 and (views != 2 or M % 256):
        raise ValueError("two views need a multiple of 128 rows each")
    sg = _lib.ConvSeg()
    sg.x1, sg.w1, sg.Ci1 = x1.data_ptr(), w1.data_ptr(), Ci1
    sg.w_view_stride, sg.w1_view_stride = w_view_stride, w1_view_stride
    sg.col_bias = col_bias.data_ptr() if col_bias is not None else None
    sg.views = views
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.Ci}+{Ci1}_N{desc.Co}_seg"
    with _prof(tag, 2.0 * M * desc.Co * desc.Ci, sz * (x0.numel() + x1.numel() + M * desc.Co * (1 + (addend is not None)))):
        check(_lib.load().sm3_conv_gather_gemm_seg(C.byref(desc), _ptr(x0), _ptr(w0), C.byref(sg), _ptr(y), _ptr(addend),
                                                   _stream()), "sm3_conv_gather_gemm_seg")


def conv_seg_act(desc, x0, w0, x1, w1, col_bias, y, mask=None, relu=True, views=1, w_view_stride=0, w1_view_stride=0):
    """y = relu?(x0 w0^T + x1 w1^T + col_bias) with the ReLU bits in `mask` (sm3_conv_seg_act): with BatchNorm-scaled banks
    (linbn_scale_banks) the whole join of a Bottleneck that has a downsample branch."""
    tdt = TORCH_DTYPE[desc.dtype]
    for t, n in ((x0, "x0"), (w0, "w0"), (x1, "x1"), (w1, "w1"), (y, "y")):
        _chk(t, tdt, n)
    _chk(col_bias, torch.float32, "col_bias"); _chk(mask, torch.uint8, "mask")
    if desc.dtype == SM3_F32:
        raise ValueError("conv_seg_act: 16-bit activation types only")
    M = desc.N * desc.Ho * desc.Wo
    if desc.ntaps != 1 or desc.Hout != desc.Ho or desc.Wout != desc.Wo or desc.Hi != desc.Ho or desc.Wi != desc.Wo:
        raise ValueError("conv_seg_act: 1x1 / stride-1 descriptor expected")
    if x0.numel() != M * desc.Ci or x1.numel() % M or y.numel() != M * desc.Co:
        raise ValueError("conv_seg_act: operand size does not match descriptor")
    Ci1 = x1.numel() // M
    if Ci1 % K_CHUNK[desc.dtype] or desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError("conv_seg_act: channel counts must be multiples of the K chunk")
    if w0.numel() < (views - 1) * w_view_stride + desc.Co * desc.w_row_stride or \
            w1.numel() < (views - 1) * w1_view_stride + desc.Co * Ci1 or col_bias.numel() < views * desc.Co:
        raise ValueError("conv_seg_act: weight bank / col_bias too small")
    if mask is not None and (not relu or mask.numel() != M * desc.Co // (16 // _sz(desc.dtype))):
        raise ValueError("conv_seg_act: mask size mismatch")
    if views > 1 and (views != 2 or M % 256):
        raise ValueError("two views need a multiple of 128 rows each")
    sg = _lib.ConvSeg()
    sg.x1, sg.w1, sg.Ci1 = x1.data_ptr(), w1.data_ptr(), Ci1
    sg.w_view_stride, sg.w1_view_stride = w_view_stride, w1_view_stride
    sg.col_bias = col_bias.data_ptr()
    sg.views = views
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.Ci}+{Ci1}_N{desc.Co}_segact"
    with _prof(tag, 2.0 * M * desc.Co * (desc.Ci + Ci1), sz * (x0.numel() + x1.numel() + M * desc.Co)):
        check(_lib.load().sm3_conv_seg_act(C.byref(desc), _ptr(x0), _ptr(w0), C.byref(sg), int(relu), _ptr(y), _ptr(mask),
                                           _stream()), "sm3_conv_seg_act")


def conv_bn_act_eval(desc, x, w, scale, shift, residual, relu, y):
    """conv + eval-mode BN (+residual) (+ReLU) in one launch (sm3_conv_bn_act_eval)."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(w, tdt, "w"); _chk(y, tdt, "y"); _chk(residual, tdt, "residual")
    _chk(scale, torch.float32, "scale"); _chk(shift, torch.float32, "shift")
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("x size does not match descriptor")
    n_out = desc.N * desc.Hout * desc.Wout * desc.Co
    if y.numel() != n_out or (residual is not None and residual.numel() != n_out):
        raise ValueError("y / residual size does not match descriptor")
    if scale.numel() < desc.Co or shift.numel() < desc.Co:
        raise ValueError("scale/shift too small")
    if desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError(f"Ci={desc.Ci} is not a multiple of {K_CHUNK[desc.dtype]}")
    M = desc.N * desc.Ho * desc.Wo
    sz = _sz(desc.dtype)
    with _prof(_conv_tag(desc), 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (min(x.numel(), M * desc.ntaps * desc.Ci) + M * desc.Co * (2 if residual is not None else 1))):
        check(_lib.load().sm3_conv_bn_act_eval(C.byref(desc), _ptr(x), _ptr(w), _ptr(scale), _ptr(shift),
                                               _ptr(residual), int(relu), _ptr(y), _stream()), "sm3_conv_bn_act_eval")


def conv_bn_eval(desc, x, w, gamma, beta, running_mean, running_var, eps, residual, relu, y):
    """conv + eval-mode BN (+residual) (+ReLU) in one launch, scale/shift derived in the epilogue (sm3_conv_bn_eval)."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(w, tdt, "w"); _chk(y, tdt, "y"); _chk(residual, tdt, "residual")
    for t, n in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, torch.float32, n)
        if t is not None and t.numel() < desc.Co:
            raise ValueError(f"{n} too small")
    if running_mean is None or running_var is None:
        raise ValueError("running statistics are required")
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("x size does not match descriptor")
    n_out = desc.N * desc.Hout * desc.Wout * desc.Co
    if y.numel() != n_out or (residual is not None and residual.numel() != n_out):
        raise ValueError("y / residual size does not match descriptor")
    if desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError(f"Ci={desc.Ci} is not a multiple of {K_CHUNK[desc.dtype]}")
    M = desc.N * desc.Ho * desc.Wo
    sz = _sz(desc.dtype)
    with _prof(_conv_tag(desc), 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (min(x.numel(), M * desc.ntaps * desc.Ci) + M * desc.Co * (2 if residual is not None else 1))):
        check(_lib.load().sm3_conv_bn_eval(C.byref(desc), _ptr(x), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                           _ptr(running_var), float(eps), _ptr(residual), int(relu), _ptr(y), _stream()),
              "sm3_conv_bn_eval")


def conv_wgrad(desc, x, dy, dw):
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(dy, tdt, "dy"); _chk(dw, torch.float32, "dw")
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("x size does not match descriptor")
    if dy.numel() != desc.N * desc.Ho * desc.Wo * desc.Co:
        raise ValueError("dy size does not match descriptor")
    if dw.numel() < desc.Co * desc.w_row_stride:
        raise ValueError("dw too small")
    M = desc.N * desc.Ho * desc.Wo
    sz = _sz(desc.dtype)
    tag = "conv_wgrad"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (x.numel() + dy.numel()) + 4 * desc.Co * desc.w_row_stride):
        check(_lib.load().sm3_conv_wgrad(C.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _stream()), "sm3_conv_wgrad")


WGRAD_SLAB_CAP = 512          # slabs a deterministic weight-gradient launch may use (sm3_conv_wgrad_det); 512: the nine-tap
                              # owner on 56 x 56 x 64 has ONE tile pair, so its slices are its workgroups (256 -> 512: 189 -> 140 us)
WGRAD_SLAB_FLOATS = 1 << 26   # ... within a workspace of at most max(this, 8 slabs) floats


def wgrad_det_cap(n):
    """Slabs of n floats the deterministic weight gradient gets: a function of n alone."""
    return max(8, min(WGRAD_SLAB_CAP, WGRAD_SLAB_FLOATS // max(n, 1)))


def conv_wgrad_det(desc, x, dy, dw, slabs, cap):
    """dw += dy^T x with a fixed-order split-K sum instead of float atomics (sm3_conv_wgrad_det): bit-reproducible.
    slabs: fp32 workspace with room for `cap` matrices [Co][taps * Ci]."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(dy, tdt, "dy"); _chk(dw, torch.float32, "dw"); _chk(slabs, torch.float32, "slabs")
    n = desc.Co * desc.w_row_stride
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("x size does not match descriptor")
    if dy.numel() != desc.N * desc.Ho * desc.Wo * desc.Co:
        raise ValueError("dy size does not match descriptor")
    if dw.numel() < n or desc.w_row_stride != desc.ntaps * desc.Ci:
        raise ValueError("dw too small / padded weight rows")
    if cap < 1 or slabs.numel() < cap * n:
        raise ValueError("conv_wgrad_det: slab buffer too small")
    M = desc.N * desc.Ho * desc.Wo
    sz = _sz(desc.dtype)
    tag = "conv_wgrad"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci, sz * (x.numel() + dy.numel()) + 4 * n):
        check(_lib.load().sm3_conv_wgrad_det(C.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(slabs), int(cap), _stream()),
              "sm3_conv_wgrad_det")


def slab_reduce(slabs, nslabs, n, out, accumulate=False):
    """out[:n] (+)= sum of nslabs slabs of n floats, fixed order (sm3_slab_reduce)."""
    _chk(slabs, torch.float32, "slabs"); _chk(out, torch.float32, "out")
    if slabs.numel() < nslabs * n or out.numel() < n or n % 4 or nslabs < 1:
        raise ValueError("slab_reduce: size mismatch")
    with _prof("slab_reduce", 0.0, 4.0 * n * (nslabs + 1)):
        check(_lib.load().sm3_slab_reduce(_ptr(slabs), int(nslabs), int(n), _ptr(out), int(bool(accumulate)), _stream()),
              "sm3_slab_reduce")


def conv_wgrad_cat(desc, x, dy, dw, dy1=None, dw1=None, views=1):
    """P[v] = dy_v^T x_v -> dw [views][Co][Ci] (and, with dy1, G[v] = dy1_v^T x_v -> dw1 [views][Co1][Ci]) in one launch
    (sm3_conv_wgrad_cat), accumulated into fp32 buffers the caller zeroed."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(dy, tdt, "dy"); _chk(dy1, tdt, "dy1"); _chk(dw, torch.float32, "dw"); _chk(dw1, torch.float32, "dw1")
    M = desc.N * desc.Ho * desc.Wo
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci or dy.numel() != M * desc.Co or M % views:
        raise ValueError("conv_wgrad_cat: operand size does not match descriptor")
    if (dy1 is None) != (dw1 is None) or (dy1 is not None and dy1.numel() % M):
        raise ValueError("conv_wgrad_cat: dy1 and dw1 go together, dy1 over the same pixels")
    Co1 = dy1.numel() // M if dy1 is not None else 0
    if (Co1 and desc.Co % 128) or dw.numel() < views * desc.Co * desc.w_row_stride or \
            (Co1 and dw1.numel() < views * Co1 * desc.w_row_stride):
        raise ValueError("conv_wgrad_cat: Co must be a multiple of 128; dw / dw1 sized [views][rows][w_row_stride]")
    sz = _sz(desc.dtype)
    tag = "conv_wgrad"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}+{Co1}_v{views}"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (x.numel() + dy.numel()) + 4 * views * (desc.Co + Co1) * desc.w_row_stride):
        check(_lib.load().sm3_conv_wgrad_cat(C.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(dy1), Co1, _ptr(dw1), views,
                                             desc.Co * desc.w_row_stride, Co1 * desc.w_row_stride, _stream()),
              "sm3_conv_wgrad_cat")


SLAB_CAP = 256  # slabs per view a plain-store split-K launch may use (sm3_conv_wgrad_slabs)


def conv_wgrad_slabs(desc, x, dy, slabs, views=1, cap=None):
    """dy_v^T x_v without atomics: every pixel slice stores its partial [Co][taps*Ci] product into a slab of its own
    (sm3_conv_wgrad_slabs).  slabs: fp32 with room for views * SLAB_CAP slabs.  Returns the slabs used per view; the slabs
    of view v are slabs[(v * used + j) * Co * taps * Ci ...]."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(dy, tdt, "dy"); _chk(slabs, torch.float32, "slabs")
    M = desc.N * desc.Ho * desc.Wo
    n = desc.Co * desc.w_row_stride
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci or dy.numel() != M * desc.Co or M % views or \
            desc.w_row_stride != desc.ntaps * desc.Ci:
        raise ValueError("conv_wgrad_slabs: operand size does not match descriptor")
    # cap: slabs per view the launch may use.  It takes part in the partition of the pixel axis, so a caller that wants a view's
    # slabs to add up to the same bits in every launch configuration passes a cap that depends on n only (engine._slab_buf) --
    # the default, derived from the buffer's size, moves with whatever else grew a shared workspace.
    room = slabs.numel() // (views * n)
    cap = min(SLAB_CAP, room) if cap is None else int(cap)
    if cap < 1 or cap > room or cap > SLAB_CAP:
        raise ValueError("conv_wgrad_slabs: slab buffer too small")
    used = C.c_int(0)
    sz = _sz(desc.dtype)
    tag = "conv_wgrad"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}_v{views}_slabs"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci, sz * (x.numel() + dy.numel())):
        check(_lib.load().sm3_conv_wgrad_slabs(C.byref(desc), _ptr(x), _ptr(dy), _ptr(slabs), cap, views, C.byref(used),
                                               _stream()), "sm3_conv_wgrad_slabs")
    return used.value


def _lin_tag(name, a, b):
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        return f"linbn_small|{name[6:]}_{a}_{b}"
    return "linbn_small"


def linbn_moments(slabs, nslabs, n, out, views=1, colsum=None, colsum_rows=0, s_out=None, p=0):
    """out[v] = sum of view v's nslabs slabs (fixed order); with colsum partial rows also s_out[v] = their sum
    (sm3_linbn_moments)."""
    _chk(slabs, torch.float32, "slabs"); _chk(out, torch.float32, "out"); _chk(colsum, torch.float32, "colsum")
    _chk(s_out, torch.float64, "s_out")
    if slabs.numel() < views * nslabs * n or out.numel() < views * n or n % 4:
        raise ValueError("linbn_moments: size mismatch")
    if colsum is not None and (s_out is None or colsum.numel() < views * colsum_rows * p or s_out.numel() < views * p):
        raise ValueError("linbn_moments: colsum / s_out size mismatch")
    with _prof(_lin_tag("linbn_moments", n, nslabs), 0.0, 4.0 * views * n * (nslabs + 1)):
        check(_lib.load().sm3_linbn_moments(_ptr(slabs), nslabs, n, _ptr(out), _ptr(colsum), colsum_rows, _ptr(s_out), p,
                                            views, _stream()), "sm3_linbn_moments")


def linbn_fwd_stats(dtype, G, w_dgrad, w_fwd, s, Tm, sums_ws, Cn, p, views=1):
    """Tm[v] = W G_v and the batch sums of x = y W^T as [views][p/32][2C] fp64 partial rows for bn_finalize(groups=p/32)
    (sm3_linbn_fwd_stats).  Returns the number of groups."""
    tdt = TORCH_DTYPE[dtype]
    _chk(G, torch.float32, "G"); _chk(w_dgrad, tdt, "w_dgrad"); _chk(w_fwd, tdt, "w_fwd"); _chk(s, torch.float64, "s")
    _chk(Tm, torch.float32, "Tm"); _chk(sums_ws, torch.float64, "sums_ws")
    if Cn % 32 or p % 32 or G.numel() < views * p * p or w_dgrad.numel() != p * Cn or w_fwd.numel() != Cn * p or \
            s.numel() < views * p or Tm.numel() < views * Cn * p or sums_ws.numel() < views * (p // 32) * 2 * Cn:
        raise ValueError("linbn_fwd_stats: size mismatch")
    with _prof(_lin_tag("linbn_fwd_stats", Cn, p), 2.0 * views * Cn * p * p, 4.0 * views * (Cn * p + p * p) + _sz(dtype) * 2 * Cn * p):
        check(_lib.load().sm3_linbn_fwd_stats(dtype, _ptr(G), _ptr(w_dgrad), _ptr(w_fwd), _ptr(s), _ptr(Tm), _ptr(sums_ws),
                                              Cn, p, views, _stream()), "sm3_linbn_fwd_stats")
    return p // 32


def linbn_fold(sums_ws, groups, n, out, views=1):
    """out[v] = sum of the `groups` partial rows of sums_ws [views][groups][n] (sm3_linbn_fold)."""
    _chk(sums_ws, torch.float64, "sums_ws"); _chk(out, torch.float64, "out")
    if sums_ws.numel() < views * groups * n or out.numel() < views * n:
        raise ValueError("linbn_fold: size mismatch")
    with _prof(_lin_tag("linbn_fold", n, groups), 0.0, 8.0 * views * n * (groups + 1)):
        check(_lib.load().sm3_linbn_fold(_ptr(sums_ws), groups, n, views, _ptr(out), _stream()), "sm3_linbn_fold")


def linbn_scale_banks(dtype, w3, scale3, shift3, out3, wd, scaled, shiftd, outd, bias, Cn, views=1):
    """out3[v] = diag(scale3[v]) w3, outd[v] = diag(scaled[v]) wd, bias[v] = shift3[v] + shiftd[v] (sm3_linbn_scale_banks)."""
    tdt = TORCH_DTYPE[dtype]
    for t in (w3, out3, wd, outd):
        _chk(t, tdt)
    for t in (scale3, shift3, scaled, shiftd, bias):
        _chk(t, torch.float32)
    if w3.numel() % Cn or wd.numel() % Cn:
        raise ValueError("linbn_scale_banks: bank size is not a multiple of C")
    K3, Kd = w3.numel() // Cn, wd.numel() // Cn
    if K3 % 8 or Kd % 8 or out3.numel() < views * Cn * K3 or outd.numel() < views * Cn * Kd or bias.numel() < views * Cn or \
            min(scale3.numel(), shift3.numel(), scaled.numel(), shiftd.numel()) < views * Cn:
        raise ValueError("linbn_scale_banks: size mismatch")
    with _prof(_lin_tag("linbn_scale_banks", Cn, K3 + Kd), 0.0, _sz(dtype) * (1 + views) * Cn * (K3 + Kd)):
        check(_lib.load().sm3_linbn_scale_banks(dtype, _ptr(w3), K3, _ptr(scale3), _ptr(shift3), _ptr(out3), _ptr(wd), Kd,
                                                _ptr(scaled), _ptr(shiftd), _ptr(outd), _ptr(bias), Cn, views, _stream()),
              "sm3_linbn_scale_banks")


def linbn_stats(dtype, P, w_fwd, mean, invstd, gamma, reduce_ws, groups, lsums, dgamma, dbeta, count, coef, Cn, p, views=1):
    """lsums[v] = (sum dz | sum dz * xhat): the first from stage A of bn_stats_reduce (reduce_ws, groups), the second from
    P = dz^T y; d(gamma), d(beta) accumulated; count > 0: coef [views][C][4] written too (sm3_linbn_stats)."""
    _chk(P, torch.float32, "P"); _chk(w_fwd, TORCH_DTYPE[dtype], "w_fwd")
    for t in (mean, invstd, gamma, dgamma, dbeta, coef):
        _chk(t, torch.float32)
    _chk(lsums, torch.float64, "lsums"); _chk(reduce_ws, torch.float64, "reduce_ws")
    if P.numel() < views * Cn * p or w_fwd.numel() != Cn * p or mean.numel() < views * Cn or invstd.numel() < views * Cn or \
            lsums.numel() < views * 2 * Cn or reduce_ws.numel() < views * groups * 2 * Cn or \
            (count > 0 and (coef is None or coef.numel() < views * 4 * Cn)):
        raise ValueError("linbn_stats: size mismatch")
    with _prof(_lin_tag("linbn_stats", Cn, p), 0.0, 4.0 * views * Cn * p):
        check(_lib.load().sm3_linbn_stats(dtype, _ptr(P), _ptr(w_fwd), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                          _ptr(reduce_ws), groups, _ptr(lsums), _ptr(dgamma), _ptr(dbeta), float(count),
                                          _ptr(coef), Cn, p, views, _stream()), "sm3_linbn_stats")


def linbn_coef(gsums, count, gamma, mean, invstd, coef, Cn, views=1):
    _chk(gsums, torch.float64)
    for t in (gamma, mean, invstd, coef):
        _chk(t, torch.float32)
    if gsums.numel() < views * 2 * Cn or mean.numel() < views * Cn or invstd.numel() < views * Cn or coef.numel() < views * 4 * Cn:
        raise ValueError("linbn_coef: size mismatch")
    check(_lib.load().sm3_linbn_coef(_ptr(gsums), float(count), _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(coef), Cn, views,
                                     _stream()), "sm3_linbn_coef")


def linbn_banks(dtype, w_dgrad, coef, wa, wbn, col_const, Cn, p, views=1):
    tdt = TORCH_DTYPE[dtype]
    _chk(w_dgrad, tdt, "w_dgrad"); _chk(wa, tdt, "wa"); _chk(wbn, tdt, "wbn")
    _chk(coef, torch.float32); _chk(col_const, torch.float32)
    if w_dgrad.numel() != p * Cn or wa.numel() < views * p * Cn or wbn.numel() < views * p * Cn or \
            col_const.numel() < views * p or coef.numel() < views * 4 * Cn:
        raise ValueError("linbn_banks: size mismatch")
    with _prof(_lin_tag("linbn_banks", Cn, p), 0.0, _sz(dtype) * p * Cn * (1 + 2 * views)):
        check(_lib.load().sm3_linbn_banks(dtype, _ptr(w_dgrad), _ptr(coef), _ptr(wa), _ptr(wbn), _ptr(col_const), Cn, p,
                                          views, _stream()), "sm3_linbn_banks")


def linbn_post(dtype, wbn, w_dgrad, hn, P, G, Tm, s, coef, dw, Cn, p, views=1):
    """hn[v] = wbn_v w_dgrad^T (= -H_v, dtype [views][p][p]);  dw += sum_v diag(a)(P_v - m1 s^T) - diag(b)(W G_v - mu s^T),
    W G_v from Tm or (Tm None) recomputed from G (sm3_linbn_post)."""
    tdt = TORCH_DTYPE[dtype]
    for t, n in ((wbn, "wbn"), (w_dgrad, "w_dgrad"), (hn, "hn")):
        _chk(t, tdt, n)
    for t in (P, G, Tm, coef, dw):
        _chk(t, torch.float32)
    _chk(s, torch.float64, "s")
    if wbn.numel() < views * p * Cn or w_dgrad.numel() != p * Cn or hn.numel() < views * p * p or \
            P.numel() < views * Cn * p or (G is None and Tm is None) or (G is not None and G.numel() < views * p * p) or \
            (Tm is not None and Tm.numel() < views * Cn * p) or s.numel() < views * p or \
            coef.numel() < views * 4 * Cn or dw.numel() != Cn * p or Cn % 128 or p % 32:
        raise ValueError("linbn_post: size mismatch")
    flops = 2.0 * views * p * p * Cn * (1 if Tm is not None else 2)
    with _prof(_lin_tag("linbn_post", Cn, p), flops, 4.0 * Cn * p * (2 + views) + _sz(dtype) * p * Cn * (1 + views)):
        check(_lib.load().sm3_linbn_post(dtype, _ptr(wbn), _ptr(w_dgrad), _ptr(hn), _ptr(P), _ptr(G), _ptr(Tm), _ptr(s),
                                         _ptr(coef), _ptr(dw), Cn, p, views, _stream()), "sm3_linbn_post")


def conv3x3_bnin_ok(desc, views=1):
    """True iff conv3x3_bnin can take this launch (the halo-resident kernel would run it: sm3_conv3x3_bnin_ok)."""
    return bool(_lib.load().sm3_conv3x3_bnin_ok(C.byref(desc), int(views)))


def conv3x3_bnin(desc, x_raw, in_scale, in_shift, act_out, mask_out, w, y, stat_partials, views=1):
    """y = conv3x3(relu(x_raw * in_scale[v] + in_shift[v])) + BatchNorm partial sums of y; the activation and its ReLU bits
    written on the side (sm3_conv3x3_bnin): bn_act + conv_gemm in one launch, bit for bit."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x_raw, tdt, "x_raw"); _chk(act_out, tdt, "act_out"); _chk(w, tdt, "w"); _chk(y, tdt, "y")
    _chk(in_scale, torch.float32, "in_scale"); _chk(in_shift, torch.float32, "in_shift")
    _chk(mask_out, torch.uint8, "mask_out"); _chk(stat_partials, torch.float32, "stat_partials")
    M = desc.N * desc.Ho * desc.Wo
    n_in = desc.N * desc.Hi * desc.Wi * desc.Ci
    if x_raw.numel() != n_in or act_out.numel() != n_in or mask_out.numel() != n_in // 8 or y.numel() != M * desc.Co or \
            in_scale.numel() < views * desc.Ci or in_shift.numel() < views * desc.Ci or w.numel() != desc.Co * desc.w_row_stride:
        raise ValueError("conv3x3_bnin: operand size does not match descriptor")
    if stat_partials is not None and stat_partials.numel() < conv_partial_rows(desc) * 2 * desc.Co:
        raise ValueError("conv3x3_bnin: partials workspace too small")
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}_s{desc.osy}_bnin"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci, sz * (2 * n_in + M * desc.Co) + n_in // 8):
        check(_lib.load().sm3_conv3x3_bnin(C.byref(desc), _ptr(x_raw), _ptr(in_scale), _ptr(in_shift), int(views),
                                           _ptr(act_out), _ptr(mask_out), _ptr(w), _ptr(y), _ptr(stat_partials), _stream()),
              "sm3_conv3x3_bnin")


def linbn_banks_post(dtype, w_dgrad, coef, wa, col_const, hn, P, G, Tm, s, dw, Cn, p, views=1):
    """linbn_banks + linbn_post in one launch (sm3_linbn_banks_post): same wa / col_const / hn / dw, bit for bit."""
    tdt = TORCH_DTYPE[dtype]
    for t, n in ((w_dgrad, "w_dgrad"), (wa, "wa"), (hn, "hn")):
        _chk(t, tdt, n)
    for t in (P, G, Tm, coef, dw, col_const):
        _chk(t, torch.float32)
    _chk(s, torch.float64, "s")
    if wa.numel() < views * p * Cn or w_dgrad.numel() != p * Cn or hn.numel() < views * p * p or \
            col_const.numel() < views * p or P.numel() < views * Cn * p or (G is None and Tm is None) or \
            (G is not None and G.numel() < views * p * p) or (Tm is not None and Tm.numel() < views * Cn * p) or \
            s.numel() < views * p or coef.numel() < views * 4 * Cn or dw.numel() != Cn * p or Cn % 128 or p % 32:
        raise ValueError("linbn_banks_post: size mismatch")
    flops = 2.0 * views * p * p * Cn * (1 if Tm is not None else 2)
    with _prof(_lin_tag("linbn_banks_post", Cn, p), flops,
               4.0 * Cn * p * (2 + views) + _sz(dtype) * p * Cn * (2 + views)):
        check(_lib.load().sm3_linbn_banks_post(dtype, _ptr(w_dgrad), _ptr(coef), _ptr(wa), _ptr(col_const), _ptr(hn), _ptr(P),
                                               _ptr(G), _ptr(Tm), _ptr(s), _ptr(dw), Cn, p, views, _stream()),
              "sm3_linbn_banks_post")


def conv_bn_act_fused(desc, x, w, scale, shift, residual, relu, y, mask=None, views=1):
    """y = relu?(conv(x, w) * scale[v] + shift[v] (+ residual)) with the ReLU bits in `mask`: conv -> train-mode BatchNorm ->
    (+identity) -> ReLU in one launch (sm3_conv_bn_act_fused); scale / shift [views][Co] from bn_finalize."""
    tdt = TORCH_DTYPE[desc.dtype]
    _chk(x, tdt, "x"); _chk(w, tdt, "w"); _chk(y, tdt, "y"); _chk(residual, tdt, "residual")
    _chk(scale, torch.float32, "scale"); _chk(shift, torch.float32, "shift"); _chk(mask, torch.uint8, "mask")
    if x.numel() != desc.N * desc.Hi * desc.Wi * desc.Ci:
        raise ValueError("x size does not match descriptor")
    n_out = desc.N * desc.Hout * desc.Wout * desc.Co
    if y.numel() != n_out or (residual is not None and residual.numel() != n_out):
        raise ValueError("y / residual size does not match descriptor")
    if scale.numel() < views * desc.Co or shift.numel() < views * desc.Co:
        raise ValueError("scale/shift too small")
    if mask is not None and (not relu or mask.numel() != n_out // (16 // _sz(desc.dtype))):
        raise ValueError("mask size mismatch")
    M = desc.N * desc.Ho * desc.Wo
    if views > 1 and (views != 2 or M % 256):
        raise ValueError("two views need a multiple of 128 rows each")
    if desc.Ci % K_CHUNK[desc.dtype]:
        raise ValueError(f"Ci={desc.Ci} is not a multiple of {K_CHUNK[desc.dtype]}")
    sz = _sz(desc.dtype)
    tag = _conv_tag(desc)
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|M{M}_K{desc.ntaps}x{desc.Ci}_N{desc.Co}_s{desc.osy}_bnact"
    with _prof(tag, 2.0 * M * desc.Co * desc.ntaps * desc.Ci,
               sz * (min(x.numel(), M * desc.ntaps * desc.Ci) + M * desc.Co * (2 if residual is not None else 1))):
        check(_lib.load().sm3_conv_bn_act_fused(C.byref(desc), _ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(residual),
                                                int(relu), _ptr(y), _ptr(mask), views, _stream()), "sm3_conv_bn_act_fused")


# ------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------
BN_REDUCE_GROUPS = 64
_bn_ws = {}


def _bn_workspace(device, Cn, views=1):
    """fp64 scratch of the two-stage statistics reduction (stream-ordered reuse: one per device AND stream)."""
    device = (device, _stream_handle() if device.type == "cuda" else 0)
    t = _bn_ws.get(device)
    need = views * BN_REDUCE_GROUPS * 2 * Cn
    if t is None or t.numel() < need:
        t = torch.empty(max(need, 2 * BN_REDUCE_GROUPS * 2 * 2048), dtype=torch.float64, device=device[0])
        _bn_ws[device] = t
    return t


def bn_reduce_groups(rows):
    return min(BN_REDUCE_GROUPS, (rows + 31) // 32)


def bn_stats_reduce(partials, rows, Cn, sums, views=1):
    """sums: fp64 [views][2C] output, or None to run stage A only -- then returns (workspace, groups) for bn_finalize.
    rows: partial rows of ONE view; partials holds `views` such blocks back to back."""
    _chk(partials, torch.float32, "partials"); _chk(sums, torch.float64, "sums")
    if partials.numel() < views * rows * 2 * Cn or (sums is not None and sums.numel() < views * 2 * Cn):
        raise ValueError("bn_stats_reduce: buffer too small")
    ws = _bn_workspace(partials.device, Cn, views)
    with _prof("bn_stats_reduce", 0.0, 4.0 * views * rows * 2 * Cn):
        check(_lib.load().sm3_bn_stats_reduce(_ptr(partials), rows, Cn, _ptr(sums), _ptr(ws), views, _stream()),
              "sm3_bn_stats_reduce")
    return ws, bn_reduce_groups(rows)


def bn_finalize(sums, count, Cn, gamma, beta, eps, momentum, running_mean, running_var, nbt, scale, shift,
                save_mean, save_invstd, groups=1, views=1):
    """count: elements per channel of ONE view; scale / shift / save_mean / save_invstd: [views][C]."""
    for t, n in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, torch.float32, n)
        if t is not None and t.numel() < Cn:
            raise ValueError(f"{n} too small")
    for t, n in ((scale, "scale"), (shift, "shift"), (save_mean, "save_mean"), (save_invstd, "save_invstd")):
        _chk(t, torch.float32, n)
        if t is not None and t.numel() < views * Cn:
            raise ValueError(f"{n} too small")
    _chk(sums, torch.float64, "sums"); _chk(nbt, torch.int64, "num_batches_tracked")
    if sums.numel() < views * groups * 2 * Cn:
        raise ValueError("bn_finalize: sums too small for groups")
    with _prof("bn_finalize", 0.0, 40.0 * Cn):
        check(_lib.load().sm3_bn_finalize(_ptr(sums), groups, views, float(count), Cn, _ptr(gamma), _ptr(beta), eps,
                                          momentum, _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(scale),
                                          _ptr(shift), _ptr(save_mean), _ptr(save_invstd), _stream()), "sm3_bn_finalize")


_bn_tickets = {}


def bn_stats_finalize(partials, rows, count, Cn, gamma, beta, eps, momentum, running_mean, running_var, nbt, scale, shift,
                      save_mean, save_invstd, views=1):
    """bn_stats_reduce (stage A) + bn_finalize as ONE launch (single rank): partials [views][rows][2][C] -> scale / shift /
    save_mean / save_invstd [views][C], running statistics updated view by view.  Same bits as the two-launch form."""
    _chk(partials, torch.float32, "partials")
    for t, n in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, torch.float32, n)
        if t is not None and t.numel() < Cn:
            raise ValueError(f"{n} too small")
    for t, n in ((scale, "scale"), (shift, "shift"), (save_mean, "save_mean"), (save_invstd, "save_invstd")):
        _chk(t, torch.float32, n)
        if t is not None and t.numel() < views * Cn:
            raise ValueError(f"{n} too small")
    _chk(nbt, torch.int64, "num_batches_tracked")
    if partials.numel() < views * rows * 2 * Cn:
        raise ValueError("bn_stats_finalize: partials too small")
    ws = _bn_workspace(partials.device, Cn, views)
    key = (partials.device, _stream_handle() if partials.device.type == "cuda" else 0)
    tk = _bn_tickets.get(key)
    if tk is None:  # zero once; every launch leaves them zero (stream-ordered reuse: one set per device AND stream)
        tk = _bn_tickets[key] = torch.zeros(256, dtype=torch.int32, device=partials.device)
    if (Cn + 31) // 32 > tk.numel():
        raise ValueError("bn_stats_finalize: more than 8192 channels")
    with _prof("bn_stats_finalize", 0.0, 4.0 * views * rows * 2 * Cn):
        check(_lib.load().sm3_bn_stats_finalize(_ptr(partials), rows, Cn, views, _ptr(ws), _ptr(tk), float(count), _ptr(gamma),
                                                _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                                _ptr(scale), _ptr(shift), _ptr(save_mean), _ptr(save_invstd), _stream()),
              "sm3_bn_stats_finalize")


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, eps, Cn, scale, shift):
    for t in (gamma, beta, running_mean, running_var, scale, shift):
        _chk(t, torch.float32)
    check(_lib.load().sm3_bn_eval_scale_shift(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps,
                                              Cn, _ptr(scale), _ptr(shift), _stream()), "sm3_bn_eval_scale_shift")


def subsample_colsum_rows(dtype, rows, Cn):
    return _lib.load().sm3_subsample_colsum_rows(rows, Cn, dtype)


def subsample_colsum(dtype, x, y, colsum, N, H, W, Cn, stride, views=1):
    """Column-sum partial rows of x [N, H, W, C] at the stride-th pixels (and, y given, those pixels as a compact tensor):
    sm3_subsample_colsum."""
    tdt = TORCH_DTYPE[dtype]
    _chk(x, tdt, "x"); _chk(y, tdt, "y"); _chk(colsum, torch.float32, "colsum")
    Hs, Ws = (H - 1) // stride + 1, (W - 1) // stride + 1
    if x.numel() != N * H * W * Cn or (y is not None and y.numel() != N * Hs * Ws * Cn) or N % views:
        raise ValueError("subsample_colsum: size mismatch")
    if colsum.numel() < views * subsample_colsum_rows(dtype, N // views * Hs * Ws, Cn) * Cn:
        raise ValueError("subsample_colsum: colsum too small")
    with _prof("subsample", 0.0, _sz(dtype) * N * Hs * Ws * Cn * (2 if y is not None else 1)):
        check(_lib.load().sm3_subsample_colsum(dtype, _ptr(x), _ptr(y), _ptr(colsum), N, H, W, Cn, stride, views, _stream()),
              "sm3_subsample_colsum")


def bn_act_colsum_rows(dtype, rows, Cn):
    return _lib.load().sm3_bn_act_colsum_rows(rows, Cn, dtype)


def bn_act(dtype, x, scale, shift, residual, relu, y, rows, Cn, out_f32=False, mask=None, views=1, colsum=None):
    """rows: rows of ONE view; tensors hold `views` row ranges back to back, scale/shift are [views][C].
    colsum: optional fp32 [views][bn_act_colsum_rows][C] -- per-block column sums of the stored outputs."""
    tdt = TORCH_DTYPE[dtype]
    _chk(x, tdt, "x"); _chk(residual, tdt, "residual"); _chk(scale, torch.float32); _chk(shift, torch.float32)
    _chk(y, torch.float32 if out_f32 else tdt, "y")
    n = views * rows * Cn
    if x.numel() != n or y.numel() != n or (residual is not None and residual.numel() != n):
        raise ValueError("bn_act: size mismatch")
    if scale.numel() < views * Cn or shift.numel() < views * Cn:
        raise ValueError("bn_act: scale/shift too small")
    _chk(mask, torch.uint8, "mask")
    if mask is not None and mask.numel() != n // (16 // _sz(dtype)):
        raise ValueError("bn_act: mask size mismatch")
    tag = "bn_act"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|rows{views * rows}_C{Cn}_res{int(residual is not None)}"
    _chk(colsum, torch.float32, "colsum")
    if colsum is not None and (out_f32 or colsum.numel() < views * bn_act_colsum_rows(dtype, rows, Cn) * Cn):
        raise ValueError("bn_act: colsum needs the storage dtype and [views][colsum_rows][C] floats")
    with _prof(tag, 0.0, _sz(dtype) * n * (2 if residual is None else 3)):
        if colsum is not None:
            check(_lib.load().sm3_bn_act_colsum(dtype, _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), int(relu),
                                                _ptr(y), _ptr(mask), _ptr(colsum), rows, Cn, views, _stream()),
                  "sm3_bn_act_colsum")
        else:
            check(_lib.load().sm3_bn_act(dtype, _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), int(relu),
                                         int(out_f32), _ptr(y), _ptr(mask), rows, Cn, views, _stream()), "sm3_bn_act")


def bn_add_bn_act(dtype, x, scale, shift, x2, scale2, shift2, relu, y, rows, Cn, mask=None, views=1):
    """y = relu?(x*scale + shift + x2*scale2 + shift2): the join of a Bottleneck whose identity is a downsample
    conv + BatchNorm, with BOTH normalisations applied in this one pass (sm3_bn_add_bn_act)."""
    tdt = TORCH_DTYPE[dtype]
    for t in (x, x2, y):
        _chk(t, tdt)
    for t in (scale, shift, scale2, shift2):
        _chk(t, torch.float32)
        if t.numel() < views * Cn:
            raise ValueError("bn_add_bn_act: scale/shift too small")
    n = views * rows * Cn
    if x.numel() != n or x2.numel() != n or y.numel() != n:
        raise ValueError("bn_add_bn_act: size mismatch")
    _chk(mask, torch.uint8, "mask")
    if mask is not None and mask.numel() != n // (16 // _sz(dtype)):
        raise ValueError("bn_add_bn_act: mask size mismatch")
    tag = "bn_act"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|rows{views * rows}_C{Cn}_res2"
    with _prof(tag, 0.0, _sz(dtype) * n * 3):
        check(_lib.load().sm3_bn_add_bn_act(dtype, _ptr(x), _ptr(scale), _ptr(shift), _ptr(x2), _ptr(scale2),
                                            _ptr(shift2), int(relu), _ptr(y), _ptr(mask), rows, Cn, views, _stream()),
              "sm3_bn_add_bn_act")


def bn_bwd_apply2(dtype, dz, count, side_a, side_b, rows, Cn, views=1):
    """One pass for two BatchNorms that received the same dz.  side = dict(x, mean, invstd, gamma, gsums, lsums, dgamma,
    dbeta, dx)."""
    tdt = TORCH_DTYPE[dtype]
    _chk(dz, tdt, "dz")
    n = views * rows * Cn
    if dz.numel() != n:
        raise ValueError("bn_bwd_apply2: dz size mismatch")
    sides = []
    for sd in (side_a, side_b):
        for k in ("x", "dx"):
            _chk(sd[k], tdt, k)
            if sd[k].numel() != n:
                raise ValueError(f"bn_bwd_apply2: {k} size mismatch")
        for k in ("mean", "invstd", "gamma", "dgamma", "dbeta"):
            _chk(sd.get(k), torch.float32, k)
        for k in ("gsums", "lsums"):
            _chk(sd.get(k), torch.float64, k)
            if sd.get(k) is not None and sd[k].numel() < views * 2 * Cn:
                raise ValueError("bn_bwd_apply2: sums too small")
        if sd["mean"].numel() < views * Cn or sd["invstd"].numel() < views * Cn:
            raise ValueError("bn_bwd_apply2: mean/invstd too small")
        a = _lib.BnApplySide()
        a.x, a.mean, a.invstd, a.dx = sd["x"].data_ptr(), sd["mean"].data_ptr(), sd["invstd"].data_ptr(), sd["dx"].data_ptr()
        a.global_sums = sd["gsums"].data_ptr()
        for k, f in (("gamma", "gamma"), ("lsums", "local_sums"), ("dgamma", "dgamma"), ("dbeta", "dbeta")):
            setattr(a, f, sd[k].data_ptr() if sd.get(k) is not None else None)
        sides.append(a)
    tag = "bn_bwd_apply"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|rows{views * rows}_C{Cn}_dual"
    with _prof(tag, 0.0, _sz(dtype) * n * 5):
        check(_lib.load().sm3_bn_bwd_apply2(dtype, _ptr(dz), float(count), C.byref(sides[0]), C.byref(sides[1]), rows, Cn,
                                            views, _stream()), "sm3_bn_bwd_apply2")


def bn_relu_maxpool_fwd(dtype, x, scale, shift, y, N, H, W, Cn, argmax=None, views=1):
    """Stem: y = maxpool3x3s2(relu(x*scale + shift)) in one pass (sm3_bn_relu_maxpool_fwd)."""
    tdt = TORCH_DTYPE[dtype]
    _chk(x, tdt, "x"); _chk(y, tdt, "y"); _chk(scale, torch.float32); _chk(shift, torch.float32); _chk(argmax, torch.uint8)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() != N * H * W * Cn or y.numel() != N * Ho * Wo * Cn or N % views:
        raise ValueError("bn_relu_maxpool_fwd: size mismatch")
    if scale.numel() < views * Cn or shift.numel() < views * Cn:
        raise ValueError("bn_relu_maxpool_fwd: scale/shift too small")
    if argmax is not None and argmax.numel() != y.numel():
        raise ValueError("bn_relu_maxpool_fwd: argmax size mismatch")
    with _prof("bn_relu_maxpool", 0.0, _sz(dtype) * (x.numel() + y.numel()) + (y.numel() if argmax is not None else 0)):
        check(_lib.load().sm3_bn_relu_maxpool_fwd(dtype, _ptr(x), _ptr(scale), _ptr(shift), _ptr(y), _ptr(argmax), N, H, W,
                                                  Cn, views, _stream()), "sm3_bn_relu_maxpool_fwd")


def maxpool_bn_bwd_partial_rows(N, H, W, views=1):
    return _lib.load().sm3_maxpool_bn_bwd_partial_rows(N, H, W, views)


def maxpool_bn_bwd(dtype, argmax, dy, x, scale, shift, mean, invstd, dz, partials, N, H, W, Cn, views=1):
    """Stem backward: maxpool gradient gather + recomputed ReLU mask + BatchNorm-backward phase 1 (sm3_maxpool_bn_bwd)."""
    tdt = TORCH_DTYPE[dtype]
    _chk(argmax, torch.uint8); _chk(dy, tdt, "dy"); _chk(x, tdt, "x"); _chk(dz, tdt, "dz"); _chk(partials, torch.float32)
    for t in (scale, shift, mean, invstd):
        _chk(t, torch.float32)
        if t.numel() < views * Cn:
            raise ValueError("maxpool_bn_bwd: per-channel vector too small")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() != N * H * W * Cn or dz.numel() != x.numel() or dy.numel() != N * Ho * Wo * Cn or \
            argmax.numel() != dy.numel() or N % views:
        raise ValueError("maxpool_bn_bwd: size mismatch")
    if partials.numel() < views * maxpool_bn_bwd_partial_rows(N, H, W, views) * 2 * Cn:
        raise ValueError("maxpool_bn_bwd: partials too small")
    with _prof("maxpool_bn_bwd", 0.0, _sz(dtype) * (2 * x.numel() + dy.numel()) + argmax.numel()):
        check(_lib.load().sm3_maxpool_bn_bwd(dtype, _ptr(argmax), _ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean),
                                             _ptr(invstd), _ptr(dz), _ptr(partials), N, H, W, Cn, views, _stream()),
              "sm3_maxpool_bn_bwd")


def bn_bwd_partial_rows(rows, Cn):
    return _lib.load().sm3_bn_bwd_partial_rows(rows, Cn)


def bn_bwd_reduce(dtype, dy, y, x, mean, invstd, dz, rows, Cn, partials, mask=None, views=1):
    """rows: rows of ONE view; partials: [views][bn_bwd_partial_rows(rows)][2][C]; mean/invstd: [views][C]."""
    tdt = TORCH_DTYPE[dtype]
    n = views * rows * Cn
    for t, nm in ((dy, "dy"), (y, "y"), (x, "x"), (dz, "dz")):
        _chk(t, tdt, nm)
        if t is not None and t.numel() != n:
            raise ValueError(f"bn_bwd_reduce: {nm} size mismatch")
    _chk(partials, torch.float32)
    if partials.numel() < views * bn_bwd_partial_rows(rows, Cn) * 2 * Cn:
        raise ValueError("bn_bwd_reduce: partials too small")
    _chk(mask, torch.uint8, "mask")
    if mask is not None and mask.numel() != n // (16 // _sz(dtype)):
        raise ValueError("bn_bwd_reduce: mask size mismatch")
    if x is not None and (mean.numel() < views * Cn or invstd.numel() < views * Cn):
        raise ValueError("bn_bwd_reduce: mean/invstd too small")
    reads = 1 + (x is not None) + (1 if (y is not None and mask is None) else 0) + (1 if dz is not None else 0)
    with _prof("bn_bwd_reduce", 0.0, _sz(dtype) * n * reads + (n // 8 if mask is not None else 0)):
        check(_lib.load().sm3_bn_bwd_reduce(dtype, _ptr(dy), _ptr(y), _ptr(mask), _ptr(x), _ptr(mean), _ptr(invstd),
                                            _ptr(dz), rows, Cn, _ptr(partials), views, _stream()), "sm3_bn_bwd_reduce")


def bn_bwd_apply(dtype, dz, x, mean, invstd, gamma, gsums, count, lsums, dgamma, dbeta, dx, rows, Cn, views=1):
    """rows / count: of ONE view; gsums / lsums: [views][2C]; mean / invstd: [views][C]."""
    tdt = TORCH_DTYPE[dtype]
    for t, n in ((dz, "dz"), (x, "x"), (dx, "dx")):
        _chk(t, tdt, n)
        if t.numel() != views * rows * Cn:
            raise ValueError(f"bn_bwd_apply: {n} size mismatch")
    _chk(gsums, torch.float64); _chk(lsums, torch.float64)
    _chk(dgamma, torch.float32); _chk(dbeta, torch.float32); _chk(gamma, torch.float32)
    if gsums.numel() < views * 2 * Cn or (lsums is not None and lsums.numel() < views * 2 * Cn):
        raise ValueError("bn_bwd_apply: sums too small")
    tag = "bn_bwd_apply"
    if _PROFILER is not None and getattr(_PROFILER, "detail", False):
        tag += f"|rows{views * rows}_C{Cn}"
    with _prof(tag, 0.0, _sz(dtype) * views * rows * Cn * 3):
        check(_lib.load().sm3_bn_bwd_apply(dtype, _ptr(dz), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                           _ptr(gsums), float(count), _ptr(lsums), _ptr(dgamma), _ptr(dbeta), _ptr(dx),
                                           rows, Cn, views, _stream()), "sm3_bn_bwd_apply")


# ------------------------------------------------------------------------------------------
# stem / pooling / weights
# ------------------------------------------------------------------------------------------
def stem_im2col(dtype, x_nchw, cols, Kpad):
    _chk(x_nchw, torch.float32, "x"); _chk(cols, TORCH_DTYPE[dtype], "cols")
    N, Cc, H, W = x_nchw.shape
    if Cc != 3:
        raise ValueError("stem expects 3 input channels")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if cols.numel() != N * Ho * Wo * Kpad:
        raise ValueError("cols size mismatch")
    with _prof("stem_im2col", 0.0, 4.0 * x_nchw.numel() + _sz(dtype) * cols.numel()):
        check(_lib.load().sm3_stem_im2col(dtype, _ptr(x_nchw), _ptr(cols), N, H, W, Kpad, _stream()), "sm3_stem_im2col")


STEM_KDIRECT = 176  # K of the direct stem's filter bank: (kh, c) groups x 8 (kw padded), 22nd group zero


def stem_partial_rows(N, H, W):
    return _lib.load().sm3_stem_partial_rows(N, H, W)


def stem_weight_prep(dtype, w_master, w_stem, only_if=None):
    _chk(w_master, torch.float32, "w_master"); _chk(w_stem, TORCH_DTYPE[dtype], "w_stem"); _chk(only_if, torch.int32, "only_if")
    if w_master.numel() != 64 * 147 or w_stem.numel() != 64 * STEM_KDIRECT:
        raise ValueError("stem_weight_prep: size mismatch")
    check(_lib.load().sm3_stem_weight_prep_if(dtype, _ptr(w_master), _ptr(w_stem), _ptr(only_if), _stream()),
          "sm3_stem_weight_prep_if")


def stem_conv_fwd(dtype, x_nchw, w_stem, y, partials=None):
    """Direct 7x7/2 stem convolution from the NCHW fp32 batch (sm3_stem_conv_fwd; bf16 / fp16 / exact f32)."""
    _chk(x_nchw, torch.float32, "x"); _chk(w_stem, TORCH_DTYPE[dtype], "w_stem"); _chk(y, TORCH_DTYPE[dtype], "y")
    _chk(partials, torch.float32, "partials")
    N, Cc, H, W = x_nchw.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if Cc != 3 or w_stem.numel() != 64 * STEM_KDIRECT or y.numel() != N * Ho * Wo * 64:
        raise ValueError("stem_conv_fwd: size mismatch")
    if partials is not None and partials.numel() < stem_partial_rows(N, H, W) * 2 * 64:
        raise ValueError("stem_conv_fwd: partials too small")
    M = N * Ho * Wo
    with _prof("stem_conv_fwd", 2.0 * M * 64 * 147, 4.0 * x_nchw.numel() + _sz(dtype) * y.numel()):
        check(_lib.load().sm3_stem_conv_fwd(dtype, _ptr(x_nchw), _ptr(w_stem), _ptr(y), _ptr(partials), N, H, W, _stream()),
              "sm3_stem_conv_fwd")


STEM_WGRAD_SLABS = 768  # SM3_STEM_WGRAD_SLABS (include/sm3_hip.h)


class StemImage:
    """The images of one encoder pass rounded to the 16-bit type once (sm3_stem_image_prep): t is [N, 3, H, Wp] `dtype` with
    the 7x7 convolution's left / right zero padding materialised; shape is the NCHW shape of the fp32 images it came from."""
    __slots__ = ("t", "shape")

    def __init__(self, t, shape):
        self.t, self.shape = t, tuple(shape)

    def record_stream(self, st):
        self.t.record_stream(st)


def stem_image_cols(W):
    return (W + 6 + 7) // 8 * 8


def stem_image_prep(dtype, views):
    """views: one or two NCHW fp32 tensors of equal shape -> StemImage of the batch `views[0]` then `views[1]` (no torch.cat)."""
    x0 = views[0]
    x1 = views[1] if len(views) > 1 else None
    _chk(x0, torch.float32, "x0"); _chk(x1, torch.float32, "x1")
    if x0.dim() != 4 or x0.shape[1] != 3 or (x1 is not None and x1.shape != x0.shape) or len(views) > 2:
        raise ValueError("stem_image_prep: NCHW fp32 views of equal shape")
    B, _, H, W = x0.shape
    V = len(views)
    out = torch.empty(V * B, 3, H, stem_image_cols(W), dtype=TORCH_DTYPE[dtype], device=x0.device)
    with _prof("stem_image_prep", 0.0, 4.0 * V * x0.numel() + _sz(dtype) * out.numel()):
        check(_lib.load().sm3_stem_image_prep(dtype, _ptr(x0), _ptr(x1), _ptr(out), B, V, H, W, _stream()),
              "sm3_stem_image_prep")
    return StemImage(out, (V * B, 3, H, W))


def stem_conv_fwd16(dtype, img, w_stem, y, partials=None):
    """sm3_stem_conv_fwd16: the direct stem on a StemImage (LDS-DMA staged 16-bit rows, no conversion in the kernel)."""
    tdt = TORCH_DTYPE[dtype]
    _chk(img.t, tdt, "ximg"); _chk(w_stem, tdt, "w_stem"); _chk(y, tdt, "y"); _chk(partials, torch.float32, "partials")
    N, Cc, H, W = img.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if Cc != 3 or w_stem.numel() != 64 * STEM_KDIRECT or y.numel() != N * Ho * Wo * 64 or \
            img.t.numel() != N * 3 * H * stem_image_cols(W):
        raise ValueError("stem_conv_fwd16: size mismatch")
    if partials is not None and partials.numel() < stem_partial_rows(N, H, W) * 2 * 64:
        raise ValueError("stem_conv_fwd16: partials too small")
    M = N * Ho * Wo
    with _prof("stem_conv_fwd", 2.0 * M * 64 * 147, _sz(dtype) * (img.t.numel() + y.numel())):
        check(_lib.load().sm3_stem_conv_fwd16(dtype, _ptr(img.t), _ptr(w_stem), _ptr(y), _ptr(partials), N, H, W, _stream()),
              "sm3_stem_conv_fwd16")


def stem_wgrad_bn16(dtype, img, dz, xo, mean, invstd, gamma, gsums, count, lsums, dgamma, dbeta, dw, views=1, slabs=None):
    """sm3_stem_wgrad_bn16: stem_wgrad_bn on a StemImage."""
    tdt = TORCH_DTYPE[dtype]
    _chk(img.t, tdt, "ximg"); _chk(dz, tdt, "dz"); _chk(xo, tdt, "xo"); _chk(dw, torch.float32, "dw")
    for t in (mean, invstd, gamma, dgamma, dbeta):
        _chk(t, torch.float32)
    _chk(gsums, torch.float64); _chk(lsums, torch.float64); _chk(slabs, torch.float32, "slabs")
    N, Cc, H, W = img.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if Cc != 3 or dz.numel() != N * Ho * Wo * 64 or xo.numel() != dz.numel() or dw.numel() < 64 * 147 or N % views or \
            img.t.numel() != N * 3 * H * stem_image_cols(W):
        raise ValueError("stem_wgrad_bn16: size mismatch")
    if mean.numel() < views * 64 or invstd.numel() < views * 64 or gsums.numel() < views * 128 or \
            (lsums is not None and lsums.numel() < views * 128):
        raise ValueError("stem_wgrad_bn16: per-channel vector too small")
    if slabs is not None and slabs.numel() < STEM_WGRAD_SLABS * 64 * 147:
        raise ValueError("stem_wgrad_bn16: slab workspace too small")
    M = N * Ho * Wo
    with _prof("stem_wgrad_bn", 2.0 * M * 64 * 147, _sz(dtype) * (img.t.numel() + 2 * dz.numel())):
        check(_lib.load().sm3_stem_wgrad_bn16(dtype, _ptr(img.t), _ptr(dz), _ptr(xo), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                              _ptr(gsums), float(count), _ptr(lsums), _ptr(dgamma), _ptr(dbeta), _ptr(dw),
                                              _ptr(slabs), N, H, W, views, _stream()), "sm3_stem_wgrad_bn16")


def stem_wgrad_bn(dtype, x_nchw, dz, xo, mean, invstd, gamma, gsums, count, lsums, dgamma, dbeta, dw, views=1, slabs=None):
    """Stem weight gradient with bn1's backward apply fused into the operand load (sm3_stem_wgrad_bn; bf16 / fp16 / exact f32).
    slabs: fp32 workspace of STEM_WGRAD_SLABS * 64 * 147 floats -> fixed-order sum instead of float atomics."""
    tdt = TORCH_DTYPE[dtype]
    _chk(x_nchw, torch.float32, "x"); _chk(dz, tdt, "dz"); _chk(xo, tdt, "xo"); _chk(dw, torch.float32, "dw")
    for t in (mean, invstd, gamma, dgamma, dbeta):
        _chk(t, torch.float32)
    _chk(gsums, torch.float64); _chk(lsums, torch.float64)
    N, Cc, H, W = x_nchw.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if Cc != 3 or dz.numel() != N * Ho * Wo * 64 or xo.numel() != dz.numel() or dw.numel() < 64 * 147 or N % views:
        raise ValueError("stem_wgrad_bn: size mismatch")
    if mean.numel() < views * 64 or invstd.numel() < views * 64 or gsums.numel() < views * 128 or \
            (lsums is not None and lsums.numel() < views * 128):
        raise ValueError("stem_wgrad_bn: per-channel vector too small")
    _chk(slabs, torch.float32, "slabs")
    if slabs is not None and slabs.numel() < STEM_WGRAD_SLABS * 64 * 147:
        raise ValueError("stem_wgrad_bn: slab workspace too small")
    M = N * Ho * Wo
    with _prof("stem_wgrad_bn", 2.0 * M * 64 * 147, 4.0 * x_nchw.numel() + 2 * _sz(dtype) * dz.numel()):
        check(_lib.load().sm3_stem_wgrad_bn(dtype, _ptr(x_nchw), _ptr(dz), _ptr(xo), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                            _ptr(gsums), float(count), _ptr(lsums), _ptr(dgamma), _ptr(dbeta), _ptr(dw),
                                            _ptr(slabs), N, H, W, views, _stream()), "sm3_stem_wgrad_bn")


def maxpool_fwd(dtype, x, y, N, H, W, Cn, argmax=None):
    _chk(x, TORCH_DTYPE[dtype]); _chk(y, TORCH_DTYPE[dtype]); _chk(argmax, torch.uint8)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() != N * H * W * Cn or y.numel() != N * Ho * Wo * Cn:
        raise ValueError("maxpool_fwd: size mismatch")
    if argmax is not None and argmax.numel() != y.numel():
        raise ValueError("maxpool_fwd: argmax size mismatch")
    with _prof("maxpool_fwd", 0.0, _sz(dtype) * (x.numel() + y.numel())):
        check(_lib.load().sm3_maxpool3x3s2_fwd(dtype, _ptr(x), _ptr(y), _ptr(argmax), N, H, W, Cn, _stream()),
              "sm3_maxpool3x3s2_fwd")


def maxpool_bwd(dtype, argmax, dy, dx, N, H, W, Cn):
    _chk(argmax, torch.uint8); _chk(dy, TORCH_DTYPE[dtype]); _chk(dx, TORCH_DTYPE[dtype])
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if dx.numel() != N * H * W * Cn or dy.numel() != N * Ho * Wo * Cn or argmax.numel() != dy.numel():
        raise ValueError("maxpool_bwd: size mismatch")
    with _prof("maxpool_bwd", 0.0, _sz(dtype) * (dx.numel() + dy.numel()) + argmax.numel()):
        check(_lib.load().sm3_maxpool3x3s2_bwd(dtype, _ptr(argmax), _ptr(dy), _ptr(dx), N, H, W, Cn, _stream()),
              "sm3_maxpool3x3s2_bwd")


def avgpool_fwd(dtype, x, feat_f32, feat_t, N, HW, Cn):
    _chk(x, TORCH_DTYPE[dtype]); _chk(feat_f32, torch.float32); _chk(feat_t, TORCH_DTYPE[dtype])
    if x.numel() != N * HW * Cn:
        raise ValueError("avgpool_fwd: size mismatch")
    for t in (feat_f32, feat_t):
        if t is not None and t.numel() != N * Cn:
            raise ValueError("avgpool_fwd: feature size mismatch")
    with _prof("avgpool", 0.0, _sz(dtype) * x.numel()):
        check(_lib.load().sm3_avgpool_fwd(dtype, _ptr(x), _ptr(feat_f32), _ptr(feat_t), N, HW, Cn, _stream()),
              "sm3_avgpool_fwd")


def avgpool_bwd(dtype, dfeat, dx, N, HW, Cn):
    _chk(dfeat, TORCH_DTYPE[dtype]); _chk(dx, TORCH_DTYPE[dtype])
    if dfeat.numel() != N * Cn or dx.numel() != N * HW * Cn:
        raise ValueError("avgpool_bwd: size mismatch")
    with _prof("avgpool", 0.0, _sz(dtype) * dx.numel()):
        check(_lib.load().sm3_avgpool_bwd(dtype, _ptr(dfeat), _ptr(dx), N, HW, Cn, _stream()), "sm3_avgpool_bwd")


def weight_prep(dtype, w, Co, taps, Ci, w_fwd, ld_fwd, w_dgrad):
    _chk(w, torch.float32, "w"); _chk(w_fwd, TORCH_DTYPE[dtype]); _chk(w_dgrad, TORCH_DTYPE[dtype])
    if w.numel() != Co * taps * Ci:
        raise ValueError("weight_prep: master size mismatch")
    if w_fwd is not None and w_fwd.numel() != Co * ld_fwd:
        raise ValueError("weight_prep: w_fwd size mismatch")
    if w_dgrad is not None and w_dgrad.numel() != Co * taps * Ci:
        raise ValueError("weight_prep: w_dgrad size mismatch")
    with _prof("weight_prep", 0.0, w.numel() * (4.0 + 2 * _sz(dtype))):
        check(_lib.load().sm3_weight_prep(dtype, _ptr(w), Co, taps, Ci, _ptr(w_fwd), ld_fwd, _ptr(w_dgrad), _stream()),
              "sm3_weight_prep")


def weight_prep_table(items, device):
    """items: list of (w_master fp32, w_fwd or None, w_dgrad or None, Co, taps, Ci, ld_fwd).  Returns the device
    table (keep it alive, together with the tensors it points to) for weight_prep_batch."""
    arr = (_lib.WPrepItem * len(items))()
    total = 0
    for a, (w, wf, wd, Co, taps, Ci, ld) in zip(arr, items):
        _chk(w, torch.float32, "w")
        if w.numel() != Co * taps * Ci or (wf is not None and wf.numel() != Co * ld) or \
                (wd is not None and wd.numel() != Co * taps * Ci) or ld < taps * Ci:
            raise ValueError("weight_prep_table: size mismatch")
        a.w, a.w_fwd, a.w_dgrad = w.data_ptr(), (wf.data_ptr() if wf is not None else None), \
            (wd.data_ptr() if wd is not None else None)
        a.Co, a.taps, a.Ci, a.ld_fwd = Co, taps, Ci, ld
        total += w.numel()
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device), len(items), total


def weight_prep_batch(dtype, table, only_if=None):
    """only_if: optional 1-element int32 device tensor; the launch does nothing when it holds 0 (see weights_changed)."""
    dev_table, n, total = table
    _chk(dev_table, torch.uint8, "table"); _chk(only_if, torch.int32, "only_if")
    with _prof("weight_prep", 0.0, total * (4.0 + 2 * _sz(dtype))):
        check(_lib.load().sm3_weight_prep_batch_if(dtype, _ptr(dev_table), n, _ptr(only_if), _stream()),
              "sm3_weight_prep_batch_if")


def weights_changed(flat, state, changed):
    """changed[0] = 1 iff the fp32 words of `flat` differ from those of the previous call with this `state` (2 x int64 on
    the device, zeros before the first call).  Device-side only: no host read."""
    _chk(flat, torch.float32, "flat"); _chk(state, torch.int64, "state"); _chk(changed, torch.int32, "changed")
    if state.numel() < 2 or changed.numel() < 1:
        raise ValueError("weights_changed: state needs 2 words, changed 1")
    with _prof("weights_hash", 0.0, 4.0 * flat.numel()):
        check(_lib.load().sm3_weights_changed(_ptr(flat), flat.numel(), _ptr(state), _ptr(changed), _stream()),
              "sm3_weights_changed")


def cast_from_f32(dtype, src, dst):
    _chk(src, torch.float32); _chk(dst, TORCH_DTYPE[dtype])
    if src.numel() != dst.numel():
        raise ValueError("cast: size mismatch")
    check(_lib.load().sm3_cast_from_f32(dtype, _ptr(src), _ptr(dst), src.numel(), _stream()), "sm3_cast_from_f32")


def cast_to_f32(dtype, src, dst):
    _chk(src, TORCH_DTYPE[dtype]); _chk(dst, torch.float32)
    if src.numel() != dst.numel():
        raise ValueError("cast: size mismatch")
    check(_lib.load().sm3_cast_to_f32(dtype, _ptr(src), _ptr(dst), src.numel(), _stream()), "sm3_cast_to_f32")


# ------------------------------------------------------------------------------------------
# NT-Xent
# ------------------------------------------------------------------------------------------
def ntxent_logits(z, temperature, zn, inv_norm, logits):
    for t in (z, zn, inv_norm, logits):
        _chk(t, torch.float32)
    R, D = z.shape
    if zn.numel() != R * D or inv_norm.numel() != R or logits.numel() != R * (R - 1):
        raise ValueError("ntxent_logits: size mismatch")
    check(_lib.load().sm3_ntxent_logits(_ptr(z), R, D, temperature, _ptr(zn), _ptr(inv_norm), _ptr(logits), _stream()),
          "sm3_ntxent_logits")


def ntxent_logits_bwd(dtype, dlogits, zn, inv_norm, temperature, dz):
    _chk(dlogits, torch.float32); _chk(zn, torch.float32); _chk(inv_norm, torch.float32); _chk(dz, TORCH_DTYPE[dtype])
    R, D = zn.shape
    if dlogits.numel() != R * (R - 1) or dz.numel() != R * D:
        raise ValueError("ntxent_logits_bwd: size mismatch")
    check(_lib.load().sm3_ntxent_logits_bwd(dtype, _ptr(dlogits), _ptr(zn), _ptr(inv_norm), R, D, temperature, _ptr(dz),
                                            _stream()), "sm3_ntxent_logits_bwd")


def ce_label0(logits, weight, loss, dlogits):
    _chk(logits, torch.float32); _chk(loss, torch.float32); _chk(dlogits, torch.float32)
    R, Cc = logits.shape
    if dlogits is not None and dlogits.numel() != logits.numel():
        raise ValueError("ce_label0: size mismatch")
    # per-row terms, summed in a fixed order by the library (the loss is a function of the logits bit for bit)
    terms = torch.empty(R, dtype=torch.float32, device=logits.device) if loss is not None else None
    check(_lib.load().sm3_ce_label0(_ptr(logits), R, Cc, weight, _ptr(terms), _ptr(loss), _ptr(dlogits), _stream()),
          "sm3_ce_label0")


def ntxent_workspace_floats(R, D):
    """sm3_ntxent_fused's workspace: normalised rows, inverse norms, per-row logsumexp, per-row loss terms."""
    return R * D + 3 * R


def ntxent_fused(dtype, z, temperature, weight, workspace, loss, dz, dz_scale=None):
    """dz_scale: optional 1-element fp32 device tensor multiplied into dz (dynamic loss scale of the fp16 mode)."""
    _chk(z, torch.float32); _chk(workspace, torch.float32); _chk(loss, torch.float32); _chk(dz, TORCH_DTYPE[dtype])
    _chk(dz_scale, torch.float32, "dz_scale")
    R, D = z.shape
    if workspace.numel() < ntxent_workspace_floats(R, D) or dz.numel() != R * D:
        raise ValueError("ntxent_fused: size mismatch")
    with _prof("ntxent_fused", 6.0 * R * R * D, 4.0 * R * D * 3):
        if dz_scale is None:
            check(_lib.load().sm3_ntxent_fused(dtype, _ptr(z), R, D, temperature, weight, _ptr(workspace), _ptr(loss),
                                               _ptr(dz), _stream()), "sm3_ntxent_fused")
        else:
            check(_lib.load().sm3_ntxent_fused_scaled(dtype, _ptr(z), R, D, temperature, weight, _ptr(dz_scale),
                                                      _ptr(workspace), _ptr(loss), _ptr(dz), _stream()),
                  "sm3_ntxent_fused_scaled")


def ntxent_batch_workspace_floats(n, R, D):
    """Workspace of ntxent_fused_batch: n per-term blocks, each padded to a multiple of 4 floats."""
    return n * ((ntxent_workspace_floats(R, D) + 3) // 4 * 4)


def ntxent_fused_batch(dtype, zs, temperature, weights, workspace, loss, dzs, dz_scale=None):
    """The NT-Xent terms of a step (equal [R, D] fp32 projections, at most 4) in three launches instead of 3 per term
    (sm3_ntxent_fused_batch): loss and every dz bit-identical to ntxent_fused called term by term.  Returns False -- nothing
    launched -- when the shapes do not qualify (the caller then makes the per-term calls)."""
    n = len(zs)
    if not (1 <= n <= 4) or len(dzs) != n or len(weights) != n:
        return False
    R, D = zs[0].shape
    if D % 4 or D > 128 or R % 2 or any(tuple(z.shape) != (R, D) for z in zs):
        return False
    per = ntxent_batch_workspace_floats(1, R, D)
    _chk(workspace, torch.float32); _chk(loss, torch.float32); _chk(dz_scale, torch.float32, "dz_scale")
    for z, dz in zip(zs, dzs):
        _chk(z, torch.float32); _chk(dz, TORCH_DTYPE[dtype])
        if dz.numel() != R * D:
            raise ValueError("ntxent_fused_batch: dz size mismatch")
    if workspace.numel() < n * per:
        raise ValueError("ntxent_fused_batch: workspace too small")
    zp = (C.c_void_p * n)(*[z.data_ptr() for z in zs])
    dp = (C.c_void_p * n)(*[d.data_ptr() for d in dzs])
    wp = (C.c_float * n)(*[float(w) for w in weights])
    with _prof("ntxent_fused", 6.0 * n * R * R * D, 4.0 * n * R * D * 3):
        check(_lib.load().sm3_ntxent_fused_batch(dtype, n, zp, R, D, temperature, wp, _ptr(dz_scale), _ptr(workspace),
                                                 _ptr(loss), dp, _stream()), "sm3_ntxent_fused_batch")
    return True


def normalize_rows(z, zn, inv_norm):
    for t in (z, zn, inv_norm):
        _chk(t, torch.float32)
    R, D = z.shape
    if zn.numel() != R * D or inv_norm.numel() != R:
        raise ValueError("normalize_rows: size mismatch")
    check(_lib.load().sm3_normalize_rows(_ptr(z), R, D, _ptr(zn), _ptr(inv_norm), _stream()), "sm3_normalize_rows")


def ntxent_rect(S, self_offset, temperature, weight, loss, dz_scale=None):
    """Local anchors x gathered candidates: loss += weight * NT-Xent rows, S <- dloss/dS in place (sm3_ntxent_rect)."""
    _chk(S, torch.float32, "S"); _chk(loss, torch.float32, "loss"); _chk(dz_scale, torch.float32, "dz_scale")
    Rl, Rg = S.shape
    if Rl % 2 or self_offset < 0 or self_offset + Rl > Rg:
        raise ValueError("ntxent_rect: bad geometry")
    terms = torch.empty(Rl, dtype=torch.float32, device=S.device)  # per-row terms, summed in a fixed order by the library
    with _prof("ntxent_rect", 0.0, 8.0 * Rl * Rg):
        check(_lib.load().sm3_ntxent_rect(_ptr(S), Rl, Rg, self_offset, temperature, weight, _ptr(dz_scale), _ptr(terms),
                                          _ptr(loss), _stream()), "sm3_ntxent_rect")


def normalize_rows_bwd(dtype, dzn_a, dzn_b, zn, inv_norm, dz):
    _chk(dzn_a, torch.float32); _chk(dzn_b, torch.float32); _chk(zn, torch.float32); _chk(inv_norm, torch.float32)
    _chk(dz, TORCH_DTYPE[dtype])
    R, D = zn.shape
    if dzn_a.numel() != R * D or (dzn_b is not None and dzn_b.numel() != R * D) or dz.numel() != R * D or inv_norm.numel() != R:
        raise ValueError("normalize_rows_bwd: size mismatch")
    check(_lib.load().sm3_normalize_rows_bwd(dtype, _ptr(dzn_a), _ptr(dzn_b), _ptr(zn), _ptr(inv_norm), R, D, _ptr(dz),
                                             _stream()), "sm3_normalize_rows_bwd")


# ------------------------------------------------------------------------------------------
# optimizer
# ------------------------------------------------------------------------------------------
def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, found_inf=None):
    for t in (p, g, m, v):
        _chk(t, torch.float32)
        if t.numel() != p.numel():
            raise ValueError("adamw: size mismatch")
    _chk(found_inf, torch.int32)
    with _prof("adamw", 0.0, 28.0 * p.numel()):
        check(_lib.load().sm3_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                    step, grad_scale, _ptr(found_inf), _stream()), "sm3_adamw")


def adamw_dynamic(p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, loss_scale, steps_taken, found_inf):
    """AdamW under device-resident dynamic loss scaling (sm3_adamw_dynamic)."""
    for t in (p, g, m, v):
        _chk(t, torch.float32)
        if t.numel() != p.numel():
            raise ValueError("adamw_dynamic: size mismatch")
    _chk(loss_scale, torch.float32); _chk(steps_taken, torch.int32); _chk(found_inf, torch.int32)
    with _prof("adamw", 0.0, 28.0 * p.numel()):
        check(_lib.load().sm3_adamw_dynamic(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps,
                                            weight_decay, grad_scale, _ptr(loss_scale), _ptr(steps_taken), _ptr(found_inf),
                                            _stream()), "sm3_adamw_dynamic")


def loss_scale_update(loss_scale, found_inf, growth_tracker, steps_taken, growth_factor=2.0, backoff_factor=0.5,
                      growth_interval=2000):
    """torch.cuda.amp.GradScaler.update() on device state (sm3_loss_scale_update)."""
    _chk(loss_scale, torch.float32); _chk(found_inf, torch.int32); _chk(growth_tracker, torch.int32)
    _chk(steps_taken, torch.int32)
    check(_lib.load().sm3_loss_scale_update(_ptr(loss_scale), _ptr(found_inf), _ptr(growth_tracker), _ptr(steps_taken),
                                            growth_factor, backoff_factor, int(growth_interval), _stream()),
          "sm3_loss_scale_update")


def ema_update(target, online, momentum):
    _chk(target, torch.float32); _chk(online, torch.float32)
    if target.numel() != online.numel():
        raise ValueError("ema_update: size mismatch")
    check(_lib.load().sm3_ema_update(_ptr(target), _ptr(online), target.numel(), float(momentum), _stream()), "sm3_ema_update")


def check_finite(g, found_inf):
    _chk(g, torch.float32); _chk(found_inf, torch.int32)
    check(_lib.load().sm3_check_finite(_ptr(g), g.numel(), _ptr(found_inf), _stream()), "sm3_check_finite")


# ------------------------------------------------------------------------------------------
# multi-label heads (inference model)
# ------------------------------------------------------------------------------------------
def token_attention(dtype, qkv, out, B, S, D, nhead):
    tdt = TORCH_DTYPE[dtype]
    _chk(qkv, tdt, "qkv"); _chk(out, tdt, "out")
    if qkv.numel() != B * S * 3 * D or out.numel() != B * S * D:
        raise ValueError("token_attention: size mismatch")
    if S > 8 or nhead < 1 or nhead > 8 or D % nhead:
        raise ValueError("token_attention: at most 8 tokens / 8 heads, D divisible by nhead")
    check(_lib.load().sm3_token_attention(dtype, _ptr(qkv), _ptr(out), B, S, D, nhead, _stream()), "sm3_token_attention")


def add_layernorm(dtype, a, b, gamma, beta, eps, out, rows, D):
    tdt = TORCH_DTYPE[dtype]
    _chk(a, tdt, "a"); _chk(b, tdt, "b"); _chk(out, tdt, "out"); _chk(gamma, torch.float32); _chk(beta, torch.float32)
    if a.numel() != rows * D or out.numel() != rows * D or (b is not None and b.numel() != rows * D):
        raise ValueError("add_layernorm: size mismatch")
    if gamma.numel() != D or beta.numel() != D or D > 1024:
        raise ValueError("add_layernorm: gamma/beta must have D <= 1024 entries")
    check(_lib.load().sm3_add_layernorm(dtype, _ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), rows, D,
                                        _stream()), "sm3_add_layernorm")


def token_heads(dtype, x, W, bias, token_of, l2_norm, out, B, S, D, T):
    _chk(x, TORCH_DTYPE[dtype], "x"); _chk(W, torch.float32, "W"); _chk(bias, torch.float32, "bias")
    _chk(token_of, torch.int32, "token_of"); _chk(out, torch.float32, "out")
    if x.numel() != B * S * D or W.numel() != T * D or bias.numel() != T or token_of.numel() != T or out.numel() != B * T:
        raise ValueError("token_heads: size mismatch")
    if S > 8:
        raise ValueError("token_heads: at most 8 tokens")
    check(_lib.load().sm3_token_heads(dtype, _ptr(x), _ptr(W), _ptr(bias), _ptr(token_of), int(bool(l2_norm)), _ptr(out),
                                      B, S, D, T, _stream()), "sm3_token_heads")
