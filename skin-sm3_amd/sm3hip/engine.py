"""SM3 pre-training engine: sequences the HIP kernels of libsm3hip.so for the hot path

    dual ResNet-50 forward/backward  ->  BN-MLP projectors  ->  in-modal + cross-modal NT-Xent

mirroring SimCLRSkinV32.forward (reference src/models/simclr.py:415-482) and the autograd backward
that tools/backbone_train.py:125 triggers.  The host side is Python (as in the reference); every
arithmetic op on the path is one of our kernels, called through the C ABI with raw device
pointers.  PyTorch supplies device memory (caching allocator), the stream and torch.distributed.

Data layout in HBM
  activations   NHWC, `dtype` (bf16 for throughput, f32 for the exact-parity mode)
  master params one flat fp32 buffer; conv weights in [Cout][kh][kw][Cin] order (= torch channels_last
                OIHW), exposed to the caller as ordinary nn.Parameters that are views into it
  gradients     one flat fp32 buffer with the same offsets (weight gradients are accumulated into it
                by the wgrad kernel, so the two views of a step simply add up)
  per-step      `dtype` copies of every filter bank in forward order and in data-gradient order
"""
import math
from collections import OrderedDict

import torch

from . import ops
from ._lib import SM3_BF16, SM3_F16, SM3_F32

import os as _os
_APPLY_OUT_OF_PLACE = _os.environ.get("SM3_BN_APPLY_OOP", "0") == "1"
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
STEM_KPAD = 192  # 7*7*3 = 147 padded to a multiple of the 128-byte K chunk for both dtypes
RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


# ------------------------------------------------------------------------------------------
# flat parameter / gradient storage
# ------------------------------------------------------------------------------------------
class ParamStore:
    """All parameters of a module in one flat fp32 buffer (64-byte aligned slots); parameters become
    views, conv weights with channels_last strides so their memory is [Cout][kh][kw][Cin]."""

    def __init__(self, module, device):
        self.module = module
        self.device = device
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for name, p in module.named_parameters():
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(p.shape)
            off += (p.numel() + 15) // 16 * 16
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self._bind()

    def _view(self, flat, name):
        shape = self.shapes[name]
        n = math.prod(shape) if shape else 1
        v = flat[self.offsets[name]: self.offsets[name] + n]
        if len(shape) == 4:
            o, i, h, w = shape
            return v.view(o, h, w, i).permute(0, 3, 1, 2)  # OIHW shape, OHWI memory
        return v.view(shape)

    def _bind(self):
        params = dict(self.module.named_parameters())
        with torch.no_grad():
            for name in self.names:
                p = params[name]
                v = self._view(self.flat_p, name)
                v.copy_(p.data.to(device=self.device, dtype=torch.float32))
                p.data = v
                p.grad = None
        self._ptrs = {n: params[n].data_ptr() for n in self.names}

    def bound(self):
        params = dict(self.module.named_parameters())
        return all(params[n].data_ptr() == self._ptrs[n] and params[n].device == self.flat_p.device
                   for n in self.names)

    def rebind_if_needed(self):
        if not self.bound():
            self._bind()

    def flat2d(self, flat, name):
        """[Cout, taps*Cin] (conv / linear) or [C] view of a slot."""
        shape = self.shapes[name]
        n = math.prod(shape)
        v = flat[self.offsets[name]: self.offsets[name] + n]
        return v.view(shape[0], -1) if len(shape) >= 2 else v

    def grad_views(self, flat=None):
        flat = self.flat_g if flat is None else flat
        return [self._view(flat, n) for n in self.names]


# ------------------------------------------------------------------------------------------
# layer units
# ------------------------------------------------------------------------------------------
class ConvUnit:
    def __init__(self, name, Ci, Co, k, stride, pad, stem=False):
        self.name, self.Ci, self.Co, self.k, self.stride, self.pad, self.stem = name, Ci, Co, k, stride, pad, stem
        self.taps = k * k
        self.w_fwd = self.w_dgrad = None
        self._fd, self._dd = {}, {}

    def alloc(self, dtype, device, need_dgrad=True, direct_stem=False):
        tdt = ops.TORCH_DTYPE[dtype]
        if self.stem:
            self.w_fwd = torch.empty(self.Co, ops.STEM_KDIRECT if direct_stem else STEM_KPAD, dtype=tdt, device=device)
        else:
            self.w_fwd = torch.empty(self.Co, self.taps * self.Ci, dtype=tdt, device=device)
            if need_dgrad:
                self.w_dgrad = torch.empty(self.Ci, self.taps, self.Co, dtype=tdt, device=device)

    def refresh(self, dtype, master2d):
        if self.stem:
            ops.weight_prep(dtype, master2d, self.Co, 1, 147, self.w_fwd, STEM_KPAD, None)
        else:
            ops.weight_prep(dtype, master2d, self.Co, self.taps, self.Ci, self.w_fwd, self.taps * self.Ci,
                            self.w_dgrad)

    def fwd_desc(self, dtype, N, H, W):
        key = (dtype, N, H, W)
        if key not in self._fd:
            if self.stem:  # GEMM over the im2col rows: N here is the row count
                d = ops.fwd_desc(dtype, N, 1, 1, STEM_KPAD, self.Co, 1, 1, 0)
            else:
                d = ops.fwd_desc(dtype, N, H, W, self.Ci, self.Co, self.k, self.stride, self.pad)
            self._fd[key] = d
        return self._fd[key]

    def wgrad_desc(self, dtype, N, H, W):
        if not self.stem:
            return self.fwd_desc(dtype, N, H, W)
        key = ("wg", dtype, N)
        if key not in self._fd:
            d = ops.fwd_desc(dtype, N, 1, 1, STEM_KPAD, self.Co, 1, 1, 0)
            d.w_row_stride = 147  # gradient rows are the unpadded [64][147] master layout
            self._fd[key] = d
        return self._fd[key]

    def compact_dgrad_desc(self, dtype, N, Hs, Ws):
        """Data gradient of a 1x1 / stride-2 convolution at the pixels it touches only: a plain GEMM over the
        [N, Hs, Ws, Co] output gradient with the transposed filter bank."""
        key = ("cdg", dtype, N, Hs, Ws)
        if key not in self._dd:
            self._dd[key] = ops.fwd_desc(dtype, N, Hs, Ws, self.Co, self.Ci, 1, 1, 0)
        return self._dd[key]

    def dgrad_descs(self, dtype, N, H, W):
        key = (dtype, N, H, W)
        if key not in self._dd:
            self._dd[key] = ops.dgrad_descs(dtype, N, H, W, self.Ci, self.Co, self.k, self.stride, self.pad)
        return self._dd[key]


class BNUnit:
    def __init__(self, name, C, affine=True):
        self.name, self.C, self.affine = name, C, affine



_LANE_POOL = {}


def lane_stream_pool(device, n):
    """n HIP streams that really run concurrently with the stream that is current now AND with each other.

    HIP multiplexes its streams onto a few hardware queues (4 by default, the null stream included): two streams that land
    on the same queue execute one after the other.  Which queue a new stream gets depends on how many streams the process
    created before (RCCL's, another engine's, the caller's) -- measured here: the second trainer created in a process ran
    11 % slower than the first and the third, its lanes serialised behind the main stream's queue.  So the lane streams are
    picked by measurement, once per device and process (every engine shares them: a stream is only an order), with
    torch.cuda._sleep as the probe: two spins on streams of one queue take twice as long as one.  SM3_STREAM_CALIBRATE=0
    falls back to the first n new streams."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _LANE_POOL.setdefault(idx, [])
    if len(pool) >= n:
        return pool[:n]
    if _os.environ.get("SM3_STREAM_CALIBRATE", "1") == "0" or not hasattr(torch.cuda, "_sleep"):
        pool.extend(torch.cuda.Stream(device=device) for _ in range(n - len(pool)))
        return pool[:n]
    import time
    main = torch.cuda.current_stream(device)
    cyc = 1_000_000

    def spin(a, b):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cyc)
        if b is not None:
            with torch.cuda.stream(b):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    spin(main, None)
    one = min(spin(main, None) for _ in range(3))
    concurrent = lambda a, b: min(spin(a, b) for _ in range(2)) < 1.5 * one
    cands = [torch.cuda.Stream(device=device) for _ in range(12)]
    rest = []
    for c in cands:
        if len(pool) >= n:
            break
        if concurrent(main, c) and all(concurrent(c, s) for s in pool):
            pool.append(c)
        else:
            rest.append(c)
    pool.extend(rest[: max(0, n - len(pool))])  # fewer independent queues than lanes: take what there is
    return pool[:n]


class Rec:
    """What one conv+BN(+act) application saves for backward."""
    __slots__ = ("cu", "bu", "N", "H", "W", "Ho", "Wo", "x_in", "xo", "mean", "invstd", "y", "relu", "mask", "V",
                 "frozen_stats", "scale", "shift", "colsum", "linbn", "gram", "Tm", "in_s")


class EncoderPlan:
    def __init__(self, prefix, block_counts=(3, 4, 6, 3)):
        self.prefix = prefix
        self.stem = ConvUnit(prefix + "conv1", 3, 64, 7, 2, 3, stem=True)
        self.stem_bn = BNUnit(prefix + "bn1", 64)
        self.blocks = []
        inpl = 64
        for li, ((planes, _, stride), nblocks) in enumerate(zip(RESNET50_LAYERS, block_counts), start=1):
            for b in range(nblocks):
                p = f"{prefix}layer{li}.{b}."
                s = stride if b == 0 else 1
                blk = {
                    "c1": ConvUnit(p + "conv1", inpl, planes, 1, 1, 0), "b1": BNUnit(p + "bn1", planes),
                    "c2": ConvUnit(p + "conv2", planes, planes, 3, s, 1), "b2": BNUnit(p + "bn2", planes),
                    "c3": ConvUnit(p + "conv3", planes, planes * 4, 1, 1, 0), "b3": BNUnit(p + "bn3", planes * 4),
                }
                if b == 0:
                    blk["cd"] = ConvUnit(p + "downsample.0", inpl, planes * 4, 1, s, 0)
                    blk["bd"] = BNUnit(p + "downsample.1", planes * 4)
                    inpl = planes * 4
                self.blocks.append(blk)
        self.out_dim = inpl

    def conv_units(self):
        yield self.stem
        for blk in self.blocks:
            for k in ("c1", "c2", "c3", "cd"):
                if k in blk:
                    yield blk[k]


class ProjectorPlan:
    def __init__(self, prefix, in_dim, proj_dim):
        self.prefix = prefix
        self.l0 = ConvUnit(prefix + "0", in_dim, in_dim, 1, 1, 0)
        self.b1 = BNUnit(prefix + "1", in_dim)
        self.l3 = ConvUnit(prefix + "3", in_dim, in_dim, 1, 1, 0)
        self.b4 = BNUnit(prefix + "4", in_dim)
        self.l6 = ConvUnit(prefix + "6", in_dim, proj_dim, 1, 1, 0)
        self.b7 = BNUnit(prefix + "7", proj_dim, affine=False)

    def conv_units(self):
        return (self.l0, self.l3, self.l6)


# ------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------
def enc_mod_out_dim(enc_mod):
    return 512 * 4


class SM3Engine:
    """Runs SimCLRSkinV3 / V32 (and a bare encoder) on the HIP kernels.

    module      the nn.Module that owns the parameters/buffers (src.models.simclr.SimCLRSkinV32 mirror)
    dtype       torch.bfloat16 (MFMA bf16, fp32 accumulate) or torch.float32 (exact-f32 MFMA)
    stat_sync   None, or a callable(t: fp64 CUDA tensor) that sums t over data-parallel ranks in place
                (SyncBatchNorm semantics, tools/backbone_train.py:510); world_size scales the counts.
    """

    def __init__(self, module, dtype=torch.bfloat16, kind="v32"):
        """kind: "v32" (SimCLRSkinV32: independent cross projectors), "v3" (SimCLRSkinV3: one shared cross
        projector), "simclr" (one SimCLR branch: encoder.* + projector.*) or "encoder" (a bare ResNet)."""
        self.module = module
        self.tdt = dtype
        self.dtype = ops.dtype_code(dtype)
        self.kind = kind
        self.store = None
        self.stat_sync = None
        # one-launch BatchNorm statistics (single rank): bit-identical, 124 launches fewer per step -- and 0.8 % SLOWER in the
        # two-lane step (62.3 vs 61.7 ms, gpurun_out r4e4): the agent-scope release fence of every block writes back an L2 that
        # the other lane's convolution keeps dirty.  Opt-in.
        self.fused_stats = _os.environ.get("SM3_BN_FUSED_STATS", "0") == "1"
        self.world_size = 1
        self.grad_ready = None  # callback(first_param_name, last_param_name) for gradient-bucket overlap
        self.fuse_bn_bwd = True  # BN-backward phase 1 inside the data-gradient epilogue (sm3_conv_dgrad_bnfuse)
        self.branches = OrderedDict()
        self.cross = None
        self.meta = None
        if kind in ("v32", "v3"):
            proj_dim = module.proj_dim
            for key in ("derm", "clinic"):
                enc_mod = getattr(module, key + "_backbone").encoder
                self.branches[key] = (EncoderPlan(f"{key}_backbone.encoder.", enc_mod.block_counts),
                                      ProjectorPlan(f"{key}_backbone.projector.", enc_mod_out_dim(enc_mod), proj_dim))
            if kind == "v3":
                cp = ProjectorPlan("cross_proj.", 2048, proj_dim)
                self.cross = (cp, cp)
            else:
                self.cross = (ProjectorPlan("cross_proj.0.", 2048, proj_dim),
                              ProjectorPlan("cross_proj.1.", 2048, proj_dim))
            if getattr(module, "meta_proj", None) is not None:  # metadata-MLP extension (src/models/simclr.py)
                self.meta = ProjectorPlan("meta_proj.", module.meta_proj[0].in_features, proj_dim)
        elif kind == "simclr":
            self.branches["main"] = (EncoderPlan("encoder.", module.encoder.block_counts),
                                     ProjectorPlan("projector.", module.encoder_out_dim, module.proj_dim))
        elif kind == "encoder":
            self.branches["main"] = (EncoderPlan("", module.block_counts), None)
        else:
            raise ValueError(kind)
        self._ws = {}
        self._allocated = False
        self._lane = "main"
        self._view = 0
        self._V = 1  # views in the batch of the encoder pass being enqueued (1, or 2 back to back)
        # Both views of a branch through the encoder as ONE batch of 2B images -- half the convolution /
        # weight-gradient / BatchNorm launches, longer K loops per weight-gradient workgroup, one SyncBN statistics
        # all-reduce per BatchNorm for both views -- with BatchNorm statistics still per view (simclr.py:58-59).
        # Needs every feature map of a view to be a multiple of 128 rows (B a multiple of 128 at 224x224).
        self.pair_views = _os.environ.get("SM3_PAIR_VIEWS", "1") != "0"
        # the two views of a branch on two more streams (BN running statistics stay ordered through events): correct
        # (GPU test suite passes with it on) but measured slower -- 2 700 vs 2 780 pairs/s -- so off: two lanes
        # already keep HBM and MFMA busy, four only add contention
        self.view_lanes = _os.environ.get("SM3_VIEW_LANES", "0") == "1"
        self._bn_ev = {}
        self._ordered_bn = False  # set while the two views of a branch run on two lanes
        self.side_wgrad = _os.environ.get("SM3_SIDE_WGRAD", "0") == "1"  # round 1: slower (2600 vs 2790 pairs/s); round 6, fixed-order sums on the side stream too: 4 631-4 651 vs 4 601-4 632 (+0.5 %, noise) on the default 4 hardware queues, 4 465 vs 4 590 with GPU_MAX_HW_QUEUES=8 -- stays opt-in
        self._side = {}
        self.two_streams = True
        self._streams, self._streams_dev = None, None
        # 7x7 stem straight from the NCHW images (csrc/stem.hip): no im2col matrix, BN-backward apply fused into the
        # stem weight gradient; 16-bit MFMA in the throughput modes, v_mfma_f32_32x32x2_f32 in the exact-f32 mode
        # (SM3_DIRECT_STEM=0: the round-1 im2col + gather-GEMM path, kept for A/B runs).
        self.direct_stem = _os.environ.get("SM3_DIRECT_STEM", "1") != "0"  # all three arithmetic modes (csrc/stem.hip)
        # BatchNorm backward by linearity for conv3 -> bn3 of every Bottleneck (csrc/linbn.hip): no bn3 backward-apply
        # pass and no backward read of conv3's output.  16-bit modes only; the exact-f32 parity mode keeps two passes.
        self.linbn = self.dtype in (SM3_BF16, SM3_F16) and _os.environ.get("SM3_LINBN", "1") != "0"
        # ... and, for the blocks without a downsample branch, the forward half of it: bn3's batch statistics from the
        # moments of conv3's input, bn3 + residual + ReLU inside conv3's epilogue -- conv3's output never reaches HBM
        self.linbn_fwd = _os.environ.get("SM3_LINBN_FWD", "1") != "0"
        # ... and the downsample conv -> BatchNorm of a stage's first block in the backward pass
        self.linbn_ds = _os.environ.get("SM3_LINBN_DS", "1") != "0"
        self.linbn_join = _os.environ.get("SM3_LINBN_JOIN", "1") != "0"
        self.linbn_merge = _os.environ.get("SM3_LINBN_MERGE", "1") != "0"  # banks + post in one launch (A/B switch)
        # bn1's apply + ReLU inside conv2's halo-resident A image (sm3_conv3x3_bnin; VERDICT r4 item 2, forward half).
        # Bit-identical to the two-pass form and MEASURED SLOWER (profiles/r05_bnin_ab.txt: 26 bn_act launches / 0.5 ms of
        # HBM-bound work less, but the 3x3 forward launches drop from 740-1 013 to 538-817 TFLOP/s -- every wave transforms
        # its pieces between their landing and the barrier, where no MFMA can overlap it; two-lane step -0.3 ... -1.1 %):
        # opt-in, SM3_CONV_BNIN=1 (2 / 3: only the 64- / 128-column launches)
        self.bnin = _os.environ.get("SM3_CONV_BNIN", "0") != "0"
        self.lane_cross = _os.environ.get("SM3_LANE_CROSS", "1") != "0"  # cross-modal projector passes inside the lanes
        if self.cross is not None and self.cross[0] is self.cross[1]:
            # SimCLRSkinV3: ONE cross projector for both modalities -- inside the lanes its parameter gradients would receive
            # the two modalities' addends in whichever order the streams run; on the main stream the order is the program's
            self.lane_cross = False
        # Weight gradients as functions of their inputs (round 6): every split-K product of the step is combined by a
        # fixed-order sum of plain-store slabs (sm3_conv_wgrad_det, sm3_stem_wgrad_bn with slabs) instead of float atomics,
        # so two runs of a training produce the same bits.  SM3_WGRAD_DET=0: the atomic forms (A/B switch).
        self.det_wgrad = _os.environ.get("SM3_WGRAD_DET", "1") != "0"
        # 16-bit modes: direct stem kernels on images rounded once per step and staged by LDS-DMA (sm3_stem_image_prep,
        # sm3_stem_conv_fwd16, sm3_stem_wgrad_bn16; bit-identical to the fp32-image kernels).  SM3_STEM16=0: the round-3 kernels.
        self.stem16 = (self.direct_stem and self.dtype in (SM3_BF16, SM3_F16)
                       and _os.environ.get("SM3_STEM16", "1") != "0")

    # ---- setup ---------------------------------------------------------------------------
    def _all_conv_units(self):
        seen = set()
        plans = [p for pair in self.branches.values() for p in pair if p is not None]
        plans += list(self.cross or ())
        if self.meta is not None:
            plans.append(self.meta)
        for plan in plans:
            for cu in plan.conv_units():
                if id(cu) not in seen:
                    seen.add(id(cu))
                    yield cu

    def prepare(self, device):
        """Bind parameters into the flat store (idempotent; re-binds after module.cuda()/to())."""
        if self.store is None or self.store.flat_p.device != device:
            self.store = ParamStore(self.module, device)
            self._allocated = False
        else:
            self.store.rebind_if_needed()
        if not self._allocated:
            for cu in self._all_conv_units():
                cu.alloc(self.dtype, device, direct_stem=self.direct_stem)
            self._allocated = True
        self.buffers = dict(self.module.named_buffers())
        for name, b in self.buffers.items():
            if b.device != device:
                raise RuntimeError(f"buffer {name} is on {b.device}, expected {device}: call module.to(device) first")

    def refresh_weights(self, defer_lanes=False):
        """fp32 master -> `dtype` filter banks (forward and data-gradient order); once per step.  The device tables of
        (master, bank) pointers are cached per master buffer (the online weights, and the momentum target's when the trainer
        swaps store.flat_p for a target forward): one table per branch encoder + in-modal projector, one for the rest.
        defer_lanes (two-lane forward): only the shared part is laid out here; each branch's banks are left to
        prep_lane(key), called at the head of that branch's lane -- the two launches then overlap instead of both lanes
        waiting for one launch over everything."""
        key = (self.store.flat_p.data_ptr(), len(self.store.names), self.dtype)
        cache = self.__dict__.setdefault("_wprep_cache", {})
        if key not in cache:
            groups = {k: ([], []) for k in list(self.branches) + [None]}
            for cu in self._all_conv_units():
                wname = cu.name + ".weight"
                if wname not in self.store.offsets:
                    continue  # projector dropped by the caller (mlc_train.py:344-346 sets them to None)
                owner = next((k for k in self.branches if k != "main" and wname.startswith(k + "_")), None)
                items, stems = groups[owner]
                m = self.store.flat2d(self.store.flat_p, wname)
                if cu.stem and self.direct_stem:
                    stems.append((m, cu.w_fwd))
                elif cu.stem:
                    items.append((m, cu.w_fwd, None, cu.Co, 1, 147, STEM_KPAD))
                else:
                    items.append((m, cu.w_fwd, cu.w_dgrad, cu.Co, cu.taps, cu.Ci, cu.taps * cu.Ci))
            dev = self.store.flat_p.device
            tables = {k: (ops.weight_prep_table(items, dev) if items else None, stems) for k, (items, stems) in groups.items()}
            cache[key] = (tables, torch.zeros(2, dtype=torch.int64, device=dev), torch.ones(1, dtype=torch.int32, device=dev))
        tables, hstate, changed = cache[key]
        # Frozen masters (linear probe, multi-label heads, inference: the same encoders forward after forward) keep their
        # banks.  Whether they changed is decided ON THE DEVICE from a hash of the flat buffer -- torch's version counters
        # miss `p.data` writes and raw-pointer kernels, a stale bank would be a silent error.  A caller that knows it just
        # rewrote the masters (the fused optimizer step) sets weights_dirty and skips the hash.
        only_if = None
        if not self.__dict__.get("weights_dirty", True) and self.__dict__.get("_wprep_key") == key:
            ops.weights_changed(self.store.flat_p, hstate, changed)
            if self.__dict__.get("_hash_tracks_banks", False):
                only_if = changed
            self._hash_tracks_banks = True  # from here on the hash state is that of the masters the banks were made from
        else:
            self._hash_tracks_banks = False  # unconditional re-layout: the remembered hash no longer describes the banks
        self._wprep_key, self.weights_dirty = key, False
        self._lane_prep = {}
        for k, (table, stems) in tables.items():
            if defer_lanes and k is not None:
                self._lane_prep[k] = (table, stems, only_if)
            else:
                self._prep(table, stems, only_if)

    def _prep(self, table, stems, only_if):
        if table is not None:
            ops.weight_prep_batch(self.dtype, table, only_if)
        for m, w in stems:
            ops.stem_weight_prep(self.dtype, m, w, only_if)

    def prep_lane(self, key):
        """The filter banks of branch `key`, on the current (= that branch's lane) stream; see refresh_weights."""
        job = self.__dict__.get("_lane_prep", {}).pop(key, None)
        if job is not None:
            self._prep(*job)

    def _work(self, key, numel, dtype=torch.float32):
        """Stream-ordered scratch; one set per execution lane (branch stream) so concurrent branches never share."""
        key = (self._lane, key)
        t = self._ws.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(numel, 1), dtype=dtype, device=self.store.flat_p.device)
            self._ws[key] = t
        return t

    # ---- two-lane execution: the derm and clinic branches are independent until the cross-modal projectors, so
    # they run on two HIP streams: while one lane is in an HBM-bound BatchNorm pass the other can be in an
    # MFMA-bound convolution, and short kernels of one lane fill the launch gaps of the other.
    def _lane_streams(self, device):
        if not self.two_streams or len(self.branches) < 2 or device.type != "cuda":
            return None
        if self._streams is None or self._streams_dev != device:
            keys = list(self.branches)
            if self.view_lanes:  # second view of every branch on a lane of its own
                keys += [k + "#1" for k in self.branches]
            self._streams = dict(zip(keys, lane_stream_pool(device, len(keys))))
            self._streams_dev = device
        return self._streams

    class _Lane:
        def __init__(self, eng, key, stream):
            self.eng, self.key, self.stream, self.ctx = eng, key, stream, None

        def __enter__(self):
            self.prev = self.eng._lane
            self.eng._lane = self.key
            self.prev_view = self.eng._view
            self.eng._view = 1 if self.key.endswith("#1") else 0
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream())  # everything enqueued so far is visible
                self.ctx = torch.cuda.stream(self.stream)
                self.ctx.__enter__()
                self.pin = ops.stream_scope()  # the lane's raw stream handle, looked up once per lane entry
                self.pin.__enter__()
            return self

        def __exit__(self, *exc):
            if self.ctx is not None:
                self.pin.__exit__(*exc)
                self.ctx.__exit__(*exc)
            self.eng._lane = self.prev
            self.eng._view = self.prev_view
            return False

    def lane(self, key, streams):
        return SM3Engine._Lane(self, key, streams[key] if streams else None)

    @staticmethod
    def _join(streams):
        """Main stream waits for every lane."""
        if streams:
            cur = torch.cuda.current_stream()
            for st in streams.values():
                cur.wait_stream(st)

    @staticmethod
    def _share(t, streams):
        """Tensor allocated on one stream, also used on others: tell the caching allocator."""
        if streams and t is not None and t.is_cuda:
            t.record_stream(torch.cuda.current_stream())
            for st in streams.values():
                t.record_stream(st)
        return t

    def _p(self, name):
        return self.store.flat2d(self.store.flat_p, name)

    def _g(self, name):
        return self.store.flat2d(self.store.flat_g, name)

    # ---- conv + BN (+residual) (+ReLU) ---------------------------------------------------
    def conv_bn(self, cu, bu, x, N, H, W, relu, residual=None, train=True, save=None, out_f32=False, y_out=None,
                apply=True, scale_shift=None, res_affine=None, pending=None, colsum=None, bn_in=None):
        """One conv + BatchNorm (+residual) (+ReLU) unit on N images.  With self._V == 2 the batch is two views back
        to back (N = 2B): one convolution launch, BatchNorm statistics / running-statistics updates per view.
        apply=False: stop after the statistics -- returns the pre-BatchNorm tensor, scale/shift are left in
        `scale_shift` for the consumer that applies them (the join of a downsample block, the stem's fused
        BN+ReLU+maxpool).  res_affine=(scale2, shift2): `residual` is such a pre-BatchNorm tensor and is normalised
        inside this unit's apply pass.
        bn_in=(x_raw, scale, shift, mask): x is still EMPTY -- it is the activation relu(x_raw * scale[v] + shift[v]) of the
        producer unit (called with apply=False), which this unit's convolution computes on its staged input image and writes
        to x / mask on the side (sm3_conv3x3_bnin: the producer's apply pass disappears).
        pending (data parallel only): a list shared by the two BatchNorms that meet at a residual join.  The unit called
        with apply=False (the downsample branch) leaves its per-rank statistic sums in the first half of a shared buffer
        and queues its finalize there instead of synchronising; the unit called next with the same list (conv3) puts
        its sums behind them, all-reduces BOTH in one collective and runs the queued finalize before its own."""
        dev = x.t.device if isinstance(x, ops.StemImage) else x.device
        direct = cu.stem and self.direct_stem  # x is the NCHW fp32 image batch (or its StemImage), N / H / W its geometry
        if direct:
            d = None
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            rows = N * Ho * Wo
        else:
            d = cu.fwd_desc(self.dtype, N, H, W)
            Ho, Wo = (1, 1) if cu.stem else (d.Ho, d.Wo)
            rows = N if cu.stem else N * Ho * Wo
        C = cu.Co
        V = self._V if train else 1
        rows_v = rows // V
        xo = torch.empty(rows, C, dtype=self.tdt, device=dev)
        scale, shift = scale_shift if scale_shift is not None else (self._work("scale", 2 * 2048),
                                                                    self._work("shift", 2 * 2048))
        gamma = self._p(bu.name + ".weight") if bu.affine else None
        beta = self._p(bu.name + ".bias") if bu.affine else None
        rm, rv = self.buffers[bu.name + ".running_mean"], self.buffers[bu.name + ".running_var"]
        track = not self.__dict__.get("_no_stat_update", False)  # momentum-target pass: batch statistics, buffers untouched
        mean = invstd = None
        if train:
            prow = ops.stem_partial_rows(N, H, W) if direct else ops.conv_partial_rows(d)
            if V > 1 and ((rows_v % 128 and not direct) or prow % V):
                raise ValueError("two views in one batch need a multiple of 128 rows per view")
            partials = self._work("partials", prow * 2 * C)
            if direct:
                (ops.stem_conv_fwd16 if isinstance(x, ops.StemImage) else ops.stem_conv_fwd)(self.dtype, x, cu.w_fwd, xo, partials)
            elif bn_in is not None:
                ops.conv3x3_bnin(d, bn_in[0], bn_in[1], bn_in[2], x, bn_in[3], cu.w_fwd, xo, partials, views=V)
            else:
                ops.conv_gemm(d, x, cu.w_fwd, xo, None, partials)
            count, groups = rows_v, 1
            mean = torch.empty(V * C, dtype=torch.float32, device=dev)
            invstd = torch.empty(V * C, dtype=torch.float32, device=dev)
            deferred = False
            if self.stat_sync is not None:
                count = rows_v * self.world_size
                n = V * 2 * C
                if pending is not None and not apply:          # downsample branch: sums parked, sync left to conv3
                    pair = self._work("sums_pair", 2 * 2 * 2 * 2048, torch.float64)
                    sums = pair[:n]
                    ops.bn_stats_reduce(partials, prow // V, C, sums, views=V)
                    deferred = True
                elif pending:                                  # conv3 of that block: one all-reduce for both units
                    pair = self._work("sums_pair", 2 * 2 * 2 * 2048, torch.float64)
                    n0 = pending[0][0]
                    sums = pair[n0:n0 + n]
                    ops.bn_stats_reduce(partials, prow // V, C, sums, views=V)
                    self.stat_sync(pair[: n0 + n])
                    for _, fin in pending:
                        fin()
                    del pending[:]
                else:
                    sums = self._work("sums", 2 * 2 * 2048, torch.float64)
                    ops.bn_stats_reduce(partials, prow // V, C, sums, views=V)
                    self.stat_sync(sums[:n])  # one all-reduce for both views
            elif self.fused_stats:
                # single rank: the whole statistics chain (row-group sums, their total, mean / var / scale / shift, running
                # statistics) is ONE launch -- the last block to arrive finalizes (sm3_bn_stats_finalize)
                sums, groups = None, 0
            else:  # ... or two: stage B of the reduction folded into bn_finalize (SM3_BN_FUSED_STATS=0)
                sums, groups = ops.bn_stats_reduce(partials, prow // V, C, None, views=V)
            ordered = self._ordered_bn and dev.type == "cuda"

            def finalize(sums=sums, groups=groups):
                if ordered and self._view == 1:
                    # running_mean/var/num_batches_tracked are updated view 0 first, then view 1, as in the reference's
                    # sequential encoder(x1); encoder(x2): the view-1 lane waits for view 0's update of THIS BatchNorm
                    torch.cuda.current_stream().wait_event(self._bn_ev[bu.name])
                if sums is None:
                    ops.bn_stats_finalize(partials, prow // V, count, C, gamma, beta, BN_EPS, BN_MOMENTUM,
                                          rm if track else None, rv if track else None,
                                          self.buffers[bu.name + ".num_batches_tracked"] if track else None,
                                          scale, shift, mean, invstd, views=V)
                else:
                    ops.bn_finalize(sums, count, C, gamma, beta, BN_EPS, BN_MOMENTUM, rm if track else None,
                                    rv if track else None,
                                    self.buffers[bu.name + ".num_batches_tracked"] if track else None,
                                    scale, shift, mean, invstd, groups=groups, views=V)
                if ordered and self._view == 0:
                    ev = self._bn_ev.get(bu.name)
                    if ev is None:
                        ev = self._bn_ev[bu.name] = torch.cuda.Event()
                    ev.record()
            if deferred:
                pending.append((V * 2 * C, finalize))
            else:
                finalize()
        elif save is None and not out_f32 and apply:
            # inference: conv + running-statistics BN (+residual) (+ReLU) in ONE launch, no pre-BN tensor in HBM
            if y_out is None:
                y_out = xo
            ops.conv_bn_eval(d, x, cu.w_fwd, gamma, beta, rm, rv, BN_EPS, residual, relu, y_out)
            return y_out, Ho, Wo
        else:
            if direct:
                (ops.stem_conv_fwd16 if isinstance(x, ops.StemImage) else ops.stem_conv_fwd)(self.dtype, x, cu.w_fwd, xo, None)
            else:
                ops.conv_gemm(d, x, cu.w_fwd, xo, None, None)
            ops.bn_eval_scale_shift(gamma, beta, rm, rv, BN_EPS, C, scale, shift)
            if save is not None:
                # eval-mode BatchNorm inside an autograd graph (module.eval() with trainable parameters): the statistics
                # are constants, so backward is dx = gamma * invstd * dz and d(gamma), d(beta) are the plain sums --
                # bn_backward runs the same kernels with the batch-statistics terms zeroed (Rec.frozen_stats)
                mean, invstd = rm.clone(), torch.rsqrt(rv + BN_EPS)
        mask = None
        if not apply:  # the consumer applies scale/shift (and the ReLU, if any)
            y_out = None
        else:
            if y_out is None:
                y_out = torch.empty(rows, C, dtype=torch.float32 if out_f32 else self.tdt, device=dev)
            if save is not None and relu:  # 1 bit per element of (y > 0): what backward needs instead of re-reading y
                mask = torch.empty(rows * C // (16 // ops._sz(self.dtype)), dtype=torch.uint8, device=dev)
            cs = None
            if res_affine is not None:
                ops.bn_add_bn_act(self.dtype, xo, scale, shift, residual, res_affine[0], res_affine[1], relu, y_out,
                                  rows_v, C, mask=mask, views=V)
            else:
                if colsum is not None and save is not None and train and not out_f32:
                    cs = colsum  # += column sums of this unit's output: first moment of the next convolution's input (linbn)
                ops.bn_act(self.dtype, xo, scale, shift, residual, relu, y_out, rows_v, C, out_f32=out_f32, mask=mask,
                           views=V, colsum=cs)
        if save is not None:
            r = Rec()
            r.cu, r.bu, r.N, r.H, r.W, r.Ho, r.Wo = cu, bu, N, H, W, Ho, Wo
            r.x_in, r.xo, r.mean, r.invstd, r.y, r.relu, r.mask = x, xo, mean, invstd, y_out, relu, mask
            r.V = V
            r.frozen_stats = not train
            r.scale = r.shift = None
            r.colsum = cs if apply else None
            r.linbn = False
            r.gram = r.Tm = None
            save.append(r)
        return (y_out if apply else xo), Ho, Wo

    def bn_backward(self, r, dy, keep_dz, fused_rows=None):
        """dy: gradient w.r.t. the unit's output (post-activation).  Masks it in place when the unit has a
        ReLU.  Returns (gradient w.r.t. the conv output, dz = masked dy).  fused_rows: dy was produced by a
        data-gradient launch that already masked it and left `fused_rows` rows of partial sums in the
        "fz_partials" workspace (conv_backward(..., fuse=r)): phase 1 is skipped."""
        C = r.cu.Co
        V = r.V
        rows = r.xo.shape[0] // V  # per view
        if fused_rows is None:
            prow = ops.bn_bwd_partial_rows(rows, C)
            bpart = self._work("partials", V * prow * 2 * C)
            ops.bn_bwd_reduce(self.dtype, dy, None, r.xo, r.mean, r.invstd, dy if r.relu else None, rows, C, bpart,
                              mask=r.mask if r.relu else None, views=V)
        else:  # [V][fused_rows][2][C], left by the data-gradient launches
            prow, bpart = fused_rows, self._ws[(self._lane, "fz_partials")]
        lsums, gsums, count = self._bn_backward_sums(r, bpart, prow)
        dxo = torch.empty_like(r.xo) if (keep_dz or _APPLY_OUT_OF_PLACE) else dy
        gamma = self._p(r.bu.name + ".weight") if r.bu.affine else None
        dgamma = self._g(r.bu.name + ".weight") if r.bu.affine else None
        dbeta = self._g(r.bu.name + ".bias") if r.bu.affine else None
        ops.bn_bwd_apply(self.dtype, dy, r.xo, r.mean, r.invstd, gamma, gsums, count, lsums, dgamma, dbeta, dxo,
                         rows, C, views=V)
        return dxo, dy

    def _bn_backward_sums(self, r, bpart, prow):
        """Partial sums [V][prow][2][C] of (dz, dz*xhat) -> (local sums, global sums, global count per channel): the
        part of BatchNorm backward between its two passes, including the SyncBN exchange."""
        C, V = r.cu.Co, r.V
        rows = r.xo.shape[0] // V
        lsums = self._work("lsums", 2 * 2 * 2048, torch.float64)
        ops.bn_stats_reduce(bpart, prow, C, lsums, views=V)
        gsums, count = lsums, rows
        if r.frozen_stats:  # eval-mode BatchNorm: no mean(dz) / mean(dz * xhat) terms in dx
            gsums = self._work("zsums", 2 * 2 * 2048, torch.float64)
            gsums.zero_()
        elif self.stat_sync is not None:
            gsums = self._work("gsums", 2 * 2 * 2048, torch.float64)
            gsums[: V * 2 * C].copy_(lsums[: V * 2 * C])
            self.stat_sync(gsums[: V * 2 * C])
            count = rows * self.world_size
        return lsums, gsums, count

    def bn_backward_join(self, r3, rd, dy, fused_rows=None):
        """The two BatchNorms of a downsample block's join (out = relu(bn3(conv3) + bn_d(conv_d)), resnet.py:164-172)
        receive the same masked gradient dz: phase 1 of both, ONE statistics exchange for both under SyncBN, and one
        apply pass that reads dz once and writes both input gradients (the downsample one in place over dz).
        Returns (d conv3 output, d downsample-conv output)."""
        C, V = r3.cu.Co, r3.V
        rows = r3.xo.shape[0] // V
        if fused_rows is None:
            prow = ops.bn_bwd_partial_rows(rows, C)
            bpart = self._work("partials", V * prow * 2 * C)
            ops.bn_bwd_reduce(self.dtype, dy, None, r3.xo, r3.mean, r3.invstd, dy, rows, C, bpart, mask=r3.mask, views=V)
        else:
            prow, bpart = fused_rows, self._ws[(self._lane, "fz_partials")]
        n = V * 2 * C
        lsums = self._work("lsums2", 2 * 2 * 2 * 2048, torch.float64)  # [bn3 | downsample][V][2C]
        ops.bn_stats_reduce(bpart, prow, C, lsums, views=V)
        prow_d = ops.bn_bwd_partial_rows(rows, C)
        dpart = self._work("partials_d", V * prow_d * 2 * C)
        ops.bn_bwd_reduce(self.dtype, dy, None, rd.xo, rd.mean, rd.invstd, None, rows, C, dpart, views=V)  # dy is dz now
        ops.bn_stats_reduce(dpart, prow_d, C, lsums[n:], views=V)
        gsums, count = lsums, rows
        if r3.frozen_stats:
            gsums = self._work("zsums2", 2 * 2 * 2 * 2048, torch.float64)
            gsums.zero_()
        elif self.stat_sync is not None:
            gsums = self._work("gsums2", 2 * 2 * 2 * 2048, torch.float64)
            gsums[: 2 * n].copy_(lsums[: 2 * n])
            self.stat_sync(gsums[: 2 * n])  # one all-reduce for the two BatchNorms (and both views)
            count = rows * self.world_size
        dx3 = torch.empty_like(r3.xo)

        def side(r, g, l, dx):
            aff = r.bu.affine
            return dict(x=r.xo, mean=r.mean, invstd=r.invstd, gamma=self._p(r.bu.name + ".weight") if aff else None,
                        gsums=g, lsums=l, dgamma=self._g(r.bu.name + ".weight") if aff else None,
                        dbeta=self._g(r.bu.name + ".bias") if aff else None, dx=dx)
        ops.bn_bwd_apply2(self.dtype, dy, count, side(r3, gsums[:n], lsums[:n], dx3),
                          side(rd, gsums[n: 2 * n], lsums[n: 2 * n], dy), rows, C, views=V)
        return dx3, dy

    def _lin_desc(self, dtype, M, K, N):
        """Descriptor of a plain [M, K] x [N, K]^T product through the gather-GEMM."""
        cache = self.__dict__.setdefault("_lin_descs", {})
        key = (dtype, M, K, N)
        if key not in cache:
            cache[key] = ops.fwd_desc(dtype, M, 1, 1, K, N, 1, 1, 0)
        return cache[key]

    def _slab_buf(self, n, V):
        """Workspace for a plain-store split-K launch (ops.conv_wgrad_slabs): up to SLAB_CAP slabs of n floats per view,
        at most 64 MB per view."""
        cap = max(1, min(ops.SLAB_CAP, (1 << 24) // n))  # a function of n alone: the partition must not depend on V
        return self._work("linbn_slabs", V * cap * n), cap

    def _lin_conv_desc(self, dtype, N, H, W, Ci, Co):
        """Descriptor of a 1x1 / stride-1 convolution Ci -> Co over an [N, H, W] map (weight-gradient-kernel launches
        that are not tied to a ConvUnit: the Gram matrix of an activation)."""
        cache = self.__dict__.setdefault("_lin_descs", {})
        key = ("conv", dtype, N, H, W, Ci, Co)
        if key not in cache:
            cache[key] = ops.fwd_desc(dtype, N, H, W, Ci, Co, 1, 1, 0)
        return cache[key]

    def conv3_bn3_fused(self, cu, bu, r2, y2, idn, N, H, W, save):
        """conv3 -> bn3 (train mode) -> + identity -> ReLU of a Bottleneck (resnet.py:162-172) without its pre-BatchNorm
        tensor: bn3's batch sums are linear / quadratic forms of the moments of y2 (r2.colsum = sum y2, r2.gram =
        y2^T y2; sm3_linbn_fwd_stats), so its scale / shift are known BEFORE conv3 runs and conv3 applies them, the residual
        and the ReLU in its own epilogue (sm3_conv_bn_act_fused)."""
        dev = y2.device
        C, p, V = cu.Co, cu.Ci, self._V
        d = cu.fwd_desc(self.dtype, N, H, W)
        rows = N * d.Ho * d.Wo
        rows_v = rows // V
        if V > 1 and rows_v % 128:
            raise ValueError("two views in one batch need a multiple of 128 rows per view")
        Tm = torch.empty(V * C * p, dtype=torch.float32, device=dev)
        groups = p // 32
        ws = self._work("linbn_fws", V * groups * 2 * C, torch.float64)
        ops.linbn_fwd_stats(self.dtype, r2.gram, cu.w_dgrad, cu.w_fwd, r2.colsum, Tm, ws, C, p, V)
        count = rows_v
        if self.stat_sync is not None:
            # SyncBatchNorm: fold the partial rows first, so that ranks exchange [V][2C] sums as the two-pass form does
            gs = self._work("linbn_fold", 2 * V * 2 * C, torch.float64)
            ops.linbn_fold(ws, groups, 2 * C, gs, views=V)
            self.stat_sync(gs[: V * 2 * C])
            ws, groups = gs, 1
            count = rows_v * self.world_size
        scale, shift = self._work("scale", 2 * 2048), self._work("shift", 2 * 2048)
        mean = torch.empty(V * C, dtype=torch.float32, device=dev)
        invstd = torch.empty(V * C, dtype=torch.float32, device=dev)
        track = not self.__dict__.get("_no_stat_update", False)
        ops.bn_finalize(ws, count, C, self._p(bu.name + ".weight") if bu.affine else None,
                        self._p(bu.name + ".bias") if bu.affine else None, BN_EPS, BN_MOMENTUM,
                        self.buffers[bu.name + ".running_mean"] if track else None,
                        self.buffers[bu.name + ".running_var"] if track else None,
                        self.buffers[bu.name + ".num_batches_tracked"] if track else None,
                        scale, shift, mean, invstd, groups=groups, views=V)
        y3 = torch.empty(rows, C, dtype=self.tdt, device=dev)
        mask = torch.empty(rows * C // (16 // ops._sz(self.dtype)), dtype=torch.uint8, device=dev)
        ops.conv_bn_act_fused(d, y2, cu.w_fwd, scale, shift, idn, True, y3, mask, views=V)
        r = Rec()
        r.cu, r.bu, r.N, r.H, r.W, r.Ho, r.Wo = cu, bu, N, H, W, d.Ho, d.Wo
        r.x_in, r.xo, r.mean, r.invstd, r.y, r.relu, r.mask = y2, None, mean, invstd, y3, True, mask
        r.V = V
        r.frozen_stats = False
        r.scale = r.shift = None
        r.colsum = r.gram = None
        r.linbn = True
        r.Tm = Tm
        save.append(r)
        return y3, d.Ho, d.Wo

    def join_fused(self, blk, r2, y2, cur, N, h, w, h2, w2, save):
        """The whole join of a Bottleneck with a downsample branch -- conv3 -> bn3, downsample conv -> its BatchNorm, add,
        ReLU (resnet.py:162-172) -- as ONE two-segment GEMM over [y2 | strided block input]: both units' batch statistics
        come from input moments (sm3_linbn_fwd_stats; data parallel: one exchange for the two), their scales go into the
        filter banks and their shifts into the column bias (sm3_linbn_scale_banks).  Neither pre-BatchNorm tensor exists;
        the compact block input and its moments are kept for the backward pass."""
        dev = y2.device
        c3, b3, cd, bd = blk["c3"], blk["b3"], blk["cd"], blk["bd"]
        C, p, Cin, V = c3.Co, c3.Ci, cd.Ci, self._V
        M = N * h2 * w2
        rows_v = M // V
        if V > 1 and rows_v % 128:
            raise ValueError("two views in one batch need a multiple of 128 rows per view")
        # moments of the (strided) block input
        crow = ops.subsample_colsum_rows(self.dtype, rows_v, Cin)
        csd = self._work("linbn_cs", V * crow * Cin)
        if cd.stride == 1:
            in_s = cur
            ops.subsample_colsum(self.dtype, cur, None, csd, N, h, w, Cin, 1, V)
        else:
            in_s = torch.empty(M, Cin, dtype=self.tdt, device=dev)
            ops.subsample_colsum(self.dtype, cur, in_s, csd, N, h, w, Cin, cd.stride, V)
        slabs, cap = self._slab_buf(Cin * Cin, V)
        ns = ops.conv_wgrad_slabs(self._lin_conv_desc(self.dtype, N, h2, w2, Cin, Cin), in_s, in_s, slabs, views=V, cap=cap)
        Gd = torch.empty(V * Cin * Cin, dtype=torch.float32, device=dev)
        sd = torch.empty(V * Cin, dtype=torch.float64, device=dev)
        ops.linbn_moments(slabs, ns, Cin * Cin, Gd, views=V, colsum=csd, colsum_rows=crow, s_out=sd, p=Cin)
        # batch statistics of both units
        g3, gd = p // 32, Cin // 32
        n3, nd = V * g3 * 2 * C, V * gd * 2 * C
        ws = self._work("linbn_fws2", n3 + nd, torch.float64)
        Tm3 = torch.empty(V * C * p, dtype=torch.float32, device=dev)
        Tmd = torch.empty(V * C * Cin, dtype=torch.float32, device=dev)
        ops.linbn_fwd_stats(self.dtype, r2.gram, c3.w_dgrad, c3.w_fwd, r2.colsum, Tm3, ws[:n3], C, p, V)
        ops.linbn_fwd_stats(self.dtype, Gd, cd.w_dgrad, cd.w_fwd, sd, Tmd, ws[n3: n3 + nd], C, Cin, V)
        count = rows_v
        units = ((b3, ws[:n3], g3, ""), (bd, ws[n3: n3 + nd], gd, "_d"))
        if self.stat_sync is not None:
            # SyncBatchNorm: partial rows folded first, then ONE exchange of [bn3 | downsample][V][2C] sums for the two units
            nf = V * 2 * C
            gs = self._work("linbn_fold", 2 * nf, torch.float64)
            ops.linbn_fold(ws[:n3], g3, 2 * C, gs[:nf], views=V)
            ops.linbn_fold(ws[n3: n3 + nd], gd, 2 * C, gs[nf: 2 * nf], views=V)
            self.stat_sync(gs[: 2 * nf])
            units = ((b3, gs[:nf], 1, ""), (bd, gs[nf: 2 * nf], 1, "_d"))
            count = rows_v * self.world_size
        track = not self.__dict__.get("_no_stat_update", False)
        out = []
        for bu, wsl, groups, tag in units:
            scale, shift = self._work("scale" + tag, 2 * 2048), self._work("shift" + tag, 2 * 2048)
            mean = torch.empty(V * C, dtype=torch.float32, device=dev)
            invstd = torch.empty(V * C, dtype=torch.float32, device=dev)
            ops.bn_finalize(wsl, count, C, self._p(bu.name + ".weight") if bu.affine else None,
                            self._p(bu.name + ".bias") if bu.affine else None, BN_EPS, BN_MOMENTUM,
                            self.buffers[bu.name + ".running_mean"] if track else None,
                            self.buffers[bu.name + ".running_var"] if track else None,
                            self.buffers[bu.name + ".num_batches_tracked"] if track else None,
                            scale, shift, mean, invstd, groups=groups, views=V)
            out.append((scale, shift, mean, invstd))
        (sc3, sh3, mean3, inv3), (scd, shd, meand, invd) = out
        w3s = self._work("linbn_w3s", V * C * p, self.tdt)
        wds = self._work("linbn_wds", V * C * Cin, self.tdt)
        bias = self._work("linbn_fbias", V * C)
        ops.linbn_scale_banks(self.dtype, c3.w_fwd, sc3, sh3, w3s, cd.w_fwd, scd, shd, wds, bias, C, V)
        y3 = torch.empty(M, C, dtype=self.tdt, device=dev)
        mask = torch.empty(M * C // (16 // ops._sz(self.dtype)), dtype=torch.uint8, device=dev)
        ops.conv_seg_act(self._lin_conv_desc(self.dtype, N, h2, w2, p, C), y2, w3s, in_s, wds, bias, y3, mask, True,
                         views=V, w_view_stride=C * p, w1_view_stride=C * Cin)
        rd, r3 = Rec(), Rec()
        for r in (rd, r3):
            r.N, r.Ho, r.Wo, r.V, r.frozen_stats, r.linbn = N, h2, w2, V, False, True
            r.scale = r.shift = r.xo = r.colsum = r.gram = r.in_s = None
        rd.cu, rd.bu, rd.H, rd.W, rd.x_in, rd.mean, rd.invstd, rd.y, rd.relu, rd.mask = cd, bd, h, w, cur, meand, invd, None, False, None
        rd.in_s, rd.gram, rd.colsum, rd.Tm = in_s, Gd, sd, Tmd
        r3.cu, r3.bu, r3.H, r3.W, r3.x_in, r3.mean, r3.invstd, r3.y, r3.relu, r3.mask = c3, b3, h2, w2, y2, mean3, inv3, y3, True, mask
        r3.Tm = Tm3
        save.append(rd)
        save.append(r3)
        return y3, h2, w2

    def _lin_unit_products(self, tag, cu, N, Hs, Ws, y_in, dz, V):
        """P = dz^T y_in [V][C][Cin] of an expanding 1x1 conv unit over compact pixels: plain-store split-K slabs of the
        weight-gradient kernel, summed in a fixed order."""
        C, p = cu.Co, cu.Ci
        slabs, cap = self._slab_buf(C * p, V)
        ns = ops.conv_wgrad_slabs(self._lin_conv_desc(self.dtype, N, Hs, Ws, p, C), y_in, dz, slabs, views=V, cap=cap)
        P = self._work("linbn_P" + tag, V * C * p)
        ops.linbn_moments(slabs, ns, C * p, P, views=V)
        return P

    def _lin_unit_finish(self, tag, cu, P, G, Tm, s, coef, V):
        """Banks diag(a) W / -diag(b) W, the constant term, -H, and the unit's weight gradient.  Returns (wa, -H, const)."""
        C, p = cu.Co, cu.Ci
        wa = self._work("linbn_wa" + tag, V * p * C, self.tdt)
        cconst = self._work("linbn_const" + tag, V * p)
        Hn = self._work("linbn_H" + tag, V * p * p, self.tdt)
        if self.linbn_merge:  # one launch; -diag(b) W stays in registers (round 5)
            ops.linbn_banks_post(self.dtype, cu.w_dgrad, coef, wa, cconst, Hn, P, G, Tm, s, self._g(cu.name + ".weight"),
                                 C, p, V)
        else:
            wbn = self._work("linbn_wbn" + tag, V * p * C, self.tdt)
            ops.linbn_banks(self.dtype, cu.w_dgrad, coef, wa, wbn, cconst, C, p, V)
            ops.linbn_post(self.dtype, wbn, cu.w_dgrad, Hn, P, G, Tm, s, coef, self._g(cu.name + ".weight"), C, p, V)
        return wa[: V * p * C], Hn[: V * p * p], cconst[: V * p]

    def conv3_backward_linbn(self, r3, r2, dz, bpart, prow, rd=None, lin_d=False):
        """Backward of conv3 -> bn3 BY LINEARITY (csrc/linbn.hip; reference: the autograd backward of
        src/models/resnet.py:162-163).  dz: the masked gradient of the block output [M, C]; bpart: its partial rows
        [V][prow][2][C] (only sum(dz) is used) as left by the producing data-gradient launch.
        No pass over bn3's input or output: the weight gradient runs on dz itself, the data gradient is one GEMM over the
        two K segments [dz | y2]; the moments of y2 (r2.colsum, r2.gram) and W G (r3.Tm) came with the forward pass.
        Accumulates d(conv3.weight), d(bn3.weight/bias).
        rd: the downsample unit of the block, whose BatchNorm received the same dz (resnet.py:164-172), with its statistics
        in the SAME SyncBN exchange as bn3's.  lin_d: it goes by linearity too -- the moments of the (strided) block input
        are taken here, its weight gradient and BatchNorm parameter gradients are accumulated, and the ingredients of its
        data gradient are returned for the caller to launch (sm3_conv_gather_gemm_seg: it joins conv1's data gradient);
        otherwise its two-pass backward runs alongside, the apply pass in place over dz once the GEMMs have read it.
        Returns (dz2 = masked gradient of bn2's output, bn2's phase-1 partial rows per view,
                 d(downsample conv output) | dict(x1, wa, hn, const) | None)."""
        cu, bu = r3.cu, r3.bu
        C, p, V = cu.Co, cu.Ci, r3.V
        M = r3.N * r3.Ho * r3.Wo
        rows = M // V
        y2 = r3.x_in
        aff = bu.affine
        gamma = self._p(bu.name + ".weight") if aff else None
        local = rows if self.stat_sync is None else 0  # single rank: the local sums are the global ones
        # 1. P = dz^T y2 [V][C][p]: the weight-gradient kernel on dz itself
        P = self._lin_unit_products("", cu, r3.N, r3.Ho, r3.Wo, y2, dz, V)
        # 2. local sums [bn3 | downsample][V][2C]: sum(dz) from the fused partial rows, sum(dz * xhat) from P; d(gamma),
        #    d(beta); single rank: the coefficients (a, b, m1, mu) too
        n = V * 2 * C
        tot = n * (2 if rd is not None else 1)
        lsums = self._work("lsums2", 2 * 2 * 2 * 2048, torch.float64)
        coef = self._work("linbn_coef", V * 4 * C)
        ws, groups = ops.bn_stats_reduce(bpart, prow, C, None, views=V)  # stage A; stage B runs inside linbn_stats
        ops.linbn_stats(self.dtype, P, cu.w_fwd, r3.mean, r3.invstd, gamma, ws, groups, lsums,
                        self._g(bu.name + ".weight") if aff else None, self._g(bu.name + ".bias") if aff else None,
                        local, coef, C, p, V)
        if lin_d:
            # the downsample unit: moments of its (strided) input, taken now; same dz, same sum(dz)
            cd, bd = rd.cu, rd.bu
            Cin = cd.Ci
            affd = bd.affine
            gamma_d = self._p(bd.name + ".weight") if affd else None
            Tmd = None
            if rd.linbn:  # the forward pass ran the join by linearity (join_fused) and kept the moments
                in_s, Gd, sd, Tmd = rd.in_s, rd.gram, rd.colsum, rd.Tm
            else:
                crow = ops.subsample_colsum_rows(self.dtype, rows, Cin)
                csd = self._work("linbn_cs", V * crow * Cin)
                if cd.stride == 1:
                    in_s = rd.x_in
                    ops.subsample_colsum(self.dtype, rd.x_in, None, csd, rd.N, rd.H, rd.W, Cin, 1, V)
                else:
                    in_s = torch.empty(M, Cin, dtype=self.tdt, device=dz.device)
                    ops.subsample_colsum(self.dtype, rd.x_in, in_s, csd, rd.N, rd.H, rd.W, Cin, cd.stride, V)
                slabs, cap = self._slab_buf(Cin * Cin, V)
                ns = ops.conv_wgrad_slabs(self._lin_conv_desc(self.dtype, rd.N, rd.Ho, rd.Wo, Cin, Cin), in_s, in_s, slabs,
                                          views=V, cap=cap)
                Gd = self._work("linbn_Gd", V * Cin * Cin)
                sd = self._work("linbn_sd", V * Cin, torch.float64)
                ops.linbn_moments(slabs, ns, Cin * Cin, Gd, views=V, colsum=csd, colsum_rows=crow, s_out=sd, p=Cin)
            Pd = self._lin_unit_products("d", cd, rd.N, rd.Ho, rd.Wo, in_s, dz, V)
            coef_d = self._work("linbn_coef_d", V * 4 * C)
            ops.linbn_stats(self.dtype, Pd, cd.w_fwd, rd.mean, rd.invstd, gamma_d, ws, groups, lsums[n: 2 * n],
                            self._g(bd.name + ".weight") if affd else None, self._g(bd.name + ".bias") if affd else None,
                            local, coef_d, C, Cin, V)
        elif rd is not None:
            prow_d = ops.bn_bwd_partial_rows(rows, C)
            dpart = self._work("partials_d", V * prow_d * 2 * C)
            ops.bn_bwd_reduce(self.dtype, dz, None, rd.xo, rd.mean, rd.invstd, None, rows, C, dpart, views=V)
            ops.bn_stats_reduce(dpart, prow_d, C, lsums[n:], views=V)
        gsums, count = lsums, rows
        if self.stat_sync is not None:
            gsums = self._work("gsums2", 2 * 2 * 2 * 2048, torch.float64)
            gsums[:tot].copy_(lsums[:tot])
            self.stat_sync(gsums[:tot])  # one all-reduce for the two BatchNorms (and both views)
            count = rows * self.world_size
            ops.linbn_coef(gsums[:n], count, gamma, r3.mean, r3.invstd, coef, C, V)
            if lin_d:
                ops.linbn_coef(gsums[n: 2 * n], count, gamma_d, rd.mean, rd.invstd, coef_d, C, V)
        # 3. diag(a) W and -diag(b) W in data-gradient order, the constant term, -H; d(conv.weight) += diag(a)(P - m1 s^T) -
        #    diag(b)(W G - mu s^T)
        wa, Hn, cconst = self._lin_unit_finish("", cu, P, r2.gram, r3.Tm, r2.colsum, coef, V)
        ds = None
        if lin_d:
            wa_d, Hn_d, cconst_d = self._lin_unit_finish("d", cd, Pd, Gd, Tmd, sd, coef_d, V)
            ds = {"x1": in_s, "wa": wa_d, "hn": Hn_d, "const": cconst_d}
        # 4. data gradient [dz | y2] x [diag(a) W ; -H]^T + const, with bn2's ReLU mask and phase 1 in the epilogue
        descs, full = cu.dgrad_descs(self.dtype, r3.N, r3.H, r3.W)
        dd = descs[0]
        dz2 = torch.empty(M, p, dtype=self.tdt, device=dz.device)
        total = ops.conv_partial_rows(dd)
        part = self._work("fz_partials", total * 2 * p)
        nrows = ops.conv_dgrad_seg_bnfuse(dd, dz, wa, y2, Hn, cconst, dz2,
                                          r2.mask if r2.relu else None, None if r2.linbn else r2.xo, r2.mean, r2.invstd,
                                          part, 0, views=V, row_offset_view1=total // V, w_view_stride=p * C,
                                          w1_view_stride=p * p)
        if rd is not None and not lin_d:  # the downsample BatchNorm's apply pass, in place over dz (its last reader: 4.)
            affd = rd.bu.affine
            ops.bn_bwd_apply(self.dtype, dz, rd.xo, rd.mean, rd.invstd, self._p(rd.bu.name + ".weight") if affd else None,
                             gsums[n: 2 * n], count, lsums[n: 2 * n], self._g(rd.bu.name + ".weight") if affd else None,
                             self._g(rd.bu.name + ".bias") if affd else None, dz, rows, C, views=V)
            ds = dz
        return dz2, nrows // V, ds

    def _wgrad(self, cu, r, dxo):
        """Weight gradient on the lane's side stream: nothing on the critical path (data gradient -> BN backward ->
        ...) depends on it, so it overlaps with those HBM-bound kernels and fills their tails."""
        side = self._side_stream()
        desc = cu.wgrad_desc(self.dtype, r.N, r.H, r.W)
        gw = self._g(cu.name + ".weight")
        if side is None:
            if self.det_wgrad and desc.w_row_stride == desc.ntaps * desc.Ci:
                # fixed-order split-K sum (plain-store slabs + sm3_slab_reduce): the gradient is a function of the inputs
                n = desc.Co * desc.w_row_stride
                cap = ops.wgrad_det_cap(n)
                ops.conv_wgrad_det(desc, r.x_in, dxo, gw, self._work("wgrad_slabs", cap * n), cap)
            else:
                ops.conv_wgrad(desc, r.x_in, dxo, gw)
            return
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)  # dxo is ready
        for t in (dxo, r.x_in):
            t.record_stream(side)  # keep the allocator from recycling them while the side stream still reads
        with torch.cuda.stream(side), ops.stream_scope():
            if self.det_wgrad and desc.w_row_stride == desc.ntaps * desc.Ci:
                n = desc.Co * desc.w_row_stride
                cap = ops.wgrad_det_cap(n)
                slabs = self._work("wgrad_slabs_side", cap * n)  # the side stream's own slabs (its launches are in order)
                ops.conv_wgrad_det(desc, r.x_in, dxo, gw, slabs, cap)
            else:
                ops.conv_wgrad(desc, r.x_in, dxo, gw)

    def _side_stream(self):
        if not self.side_wgrad or self.store.flat_p.device.type != "cuda":
            return None
        st = self._side.get(self._lane)
        if st is None:
            # behind the lane streams in the measured pool (needs GPU_MAX_HW_QUEUES >= 6 to get queues of their own)
            dev = self.store.flat_p.device
            nl = len(self._streams or {}) or len(self.branches)
            st = lane_stream_pool(dev, nl + len(self._side) + 1)[nl + len(self._side)]
            self._side[self._lane] = st
        return st

    def _sync_side(self):
        """Current lane waits for its weight-gradient stream (before gradients are declared final)."""
        st = self._side.get(self._lane)
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)

    def conv_backward(self, r, dxo, need_dx=True, addend=None, into=None, fuse=None, addend_sparse=None):
        """Weight gradient (accumulated into the flat gradient buffer) and, if need_dx, the data gradient.
        fuse: the Rec of the conv+BN unit whose OUTPUT this data gradient is the gradient of; its BN-backward
        phase 1 (ReLU mask + partial sums) then runs inside the data-gradient epilogue.
        Returns (dx, fused_rows) -- fused_rows is None when nothing was fused."""
        cu = r.cu
        self._wgrad(cu, r, dxo)
        if not need_dx:
            return None, None
        descs, full = cu.dgrad_descs(self.dtype, r.N, r.H, r.W)
        if into is not None:  # accumulate into an existing gradient
            for dd in descs:
                ops.conv_gemm(dd, dxo, cu.w_dgrad, into, into, None)
            return into, None
        if full:
            dx = torch.empty(r.N * r.H * r.W, cu.Ci, dtype=self.tdt, device=dxo.device)
            V = fuse.V if fuse is not None else 1
            if fuse is not None and self.fuse_bn_bwd and (
                    V == 1 or all((dd.N * dd.Ho * dd.Wo) % 256 == 0 for dd in descs)):
                total = sum(ops.conv_partial_rows(dd) for dd in descs)
                part = self._work("fz_partials", total * 2 * cu.Ci)  # [V][total / V][2][Ci]
                per_view = total // V
                off = 0
                for dd in descs:
                    # a unit whose backward goes by linearity (Rec.linbn) needs no sum(dz * xhat) from here: its
                    # pre-BatchNorm tensor is not read
                    n = ops.conv_dgrad_bnfuse(dd, dxo, cu.w_dgrad, dx, addend, fuse.mask if fuse.relu else None,
                                              None if fuse.linbn else fuse.xo, fuse.mean, fuse.invstd, part, off, views=V,
                                              row_offset_view1=per_view + off, addend_sparse=addend_sparse)
                    off += n // V
                return dx, off
            if addend_sparse is not None:
                raise RuntimeError("a compact addend needs the fused data-gradient epilogue")
            for dd in descs:
                ops.conv_gemm(dd, dxo, cu.w_dgrad, dx, addend, None)
            return dx, None
        dx = addend.clone() if addend is not None else torch.zeros(r.N * r.H * r.W, cu.Ci, dtype=self.tdt,
                                                                   device=dxo.device)
        for dd in descs:
            ops.conv_gemm(dd, dxo, cu.w_dgrad, dx, dx, None)
        return dx, None

    # ---- encoder -------------------------------------------------------------------------
    @staticmethod
    def pair_ok(n_view, H, W):
        """Two views can share a batch when every feature map of one view is a multiple of 128 rows (the row tile of
        the convolution kernels: a tile, its BatchNorm partial row and its fused BN-backward statistics then belong
        to exactly one view).  The smallest map is the last stage's, (H/32) x (W/32) per image."""
        h, w = H, W
        for _ in range(5):  # stem, maxpool, layer2..4 halve the map
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            if (n_view * h * w) % 128:
                return False
        return True

    def encoder_forward(self, plan, x, train, feat_f32, feat_t, save=None, views=1):
        """x: NCHW fp32 [N,3,H,W] (as the loader delivers it, tools/backbone_train.py:89-92).
        Writes the pooled features into feat_f32 [N,2048] (fp32) and feat_t (dtype copy, optional).
        views=2: x holds two views back to back (N = 2B), BatchNorm statistics per view."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]  # the views of the batch, back to back
        if any(t.dtype != torch.float32 or t.dim() != 4 or t.shape[1] != 3 or t.shape != xs[0].shape for t in xs):
            raise ValueError("encoder input must be NCHW float32 with 3 channels")
        if len(xs) > 1 and not self.stem16:
            x = torch.cat(xs, 0)
        elif len(xs) == 1:
            x = xs[0]
        self._V = views if train else 1
        try:
            self._encoder_forward(plan, x, train, feat_f32, feat_t, save)
        finally:
            self._V = 1

    def _encoder_forward(self, plan, x, train, feat_f32, feat_t, save):
        if self.stem16:
            # 16-bit modes: the images are rounded ONCE (the rounding the stem kernels used to apply per staged tile, forward
            # and again in the weight gradient) into a row-padded 16-bit copy that both kernels stage by LDS-DMA; the views
            # of a pair batch are read from their own tensors -- no torch.cat of the fp32 images
            xs = [t.contiguous() for t in (x if isinstance(x, (list, tuple)) else [x])]
            x = ops.stem_image_prep(self.dtype, xs)
            dev0 = xs[0].device
        else:
            x = x.contiguous()
            dev0 = x.device
        N, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        recs = [] if save is not None else None
        if self.direct_stem:
            stem_in, sN, sH, sW = x, N, H, W  # the 7x7 convolution reads the NCHW images directly
        else:
            cols = torch.empty(N * Ho * Wo, STEM_KPAD, dtype=self.tdt, device=dev0)
            ops.stem_im2col(self.dtype, x, cols, STEM_KPAD)
            stem_in, sN, sH, sW = cols, N * Ho * Wo, 1, 1
            del cols
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        p = torch.empty(N * Hp * Wp, 64, dtype=self.tdt, device=dev0)
        amax = torch.empty(N * Hp * Wp * 64, dtype=torch.uint8, device=dev0) if save is not None else None
        lazy = train or save is not None  # not the single-launch conv+evalBN inference path
        if lazy or self.direct_stem:
            # stem BatchNorm + ReLU + max-pool in ONE pass over the pre-BN stem output: the post-ReLU map (the largest
            # activation of the network) and its ReLU mask are never stored; backward recomputes the mask
            V = self._V if train else 1
            sc = torch.empty(V * 64, dtype=torch.float32, device=dev0)
            sh = torch.empty(V * 64, dtype=torch.float32, device=dev0)
            xo, _, _ = self.conv_bn(plan.stem, plan.stem_bn, stem_in, sN, sH, sW, True, None, train, recs,
                                    apply=False, scale_shift=(sc, sh))
            if recs is not None:
                recs[0].scale, recs[0].shift = sc, sh
            ops.bn_relu_maxpool_fwd(self.dtype, xo, sc, sh, p, N, Ho, Wo, 64, amax, views=V)
            del xo
        else:
            y, _, _ = self.conv_bn(plan.stem, plan.stem_bn, stem_in, sN, sH, sW, True, None, train, recs)
            ops.maxpool_fwd(self.dtype, y, p, N, Ho, Wo, 64, amax)
        del stem_in
        cur, h, w = p, Hp, Wp
        block_recs = []
        # BatchNorm by linearity for conv3 -> bn3 (csrc/linbn.hip) needs two moments of conv3's input y2 per view: sum(y2),
        # which bn2's apply pass adds up on the side (per-block partial rows), and the Gram matrix y2^T y2, one launch of the
        # weight-gradient kernel on y2 alone (plain-store split-K slabs); sm3_linbn_moments adds both up in a fixed order,
        # so the forward pass stays bit-reproducible.
        Vt = self._V if train else 1
        lin_ok = [self.linbn and train and save is not None and b["c3"].Co % 128 == 0 and b["c3"].Ci % 64 == 0
                  for b in plan.blocks]
        for bi, (blk, lin) in enumerate(zip(plan.blocks, lin_ok)):
            br = [] if save is not None else None
            # conv1 -> bn1 -> relu -> conv2 (resnet.py:144-150).  Where conv2 runs on the halo-resident kernel, bn1's apply +
            # ReLU happens on conv2's staged input image: conv1 stops after its statistics, conv2 reads conv1's RAW output
            # and writes the activation + ReLU bits that its weight gradient and bn1's backward need on the side
            bn_in = None
            if self.bnin and train and save is not None and self.dtype != ops.SM3_F32 and blk["c2"].stride == 1:
                d1 = blk["c1"].fwd_desc(self.dtype, N, h, w)
                if ops.conv3x3_bnin_ok(blk["c2"].fwd_desc(self.dtype, N, d1.Ho, d1.Wo), Vt):
                    C1 = blk["c1"].Co
                    sc1 = torch.empty(Vt * C1, dtype=torch.float32, device=dev0)
                    sh1 = torch.empty(Vt * C1, dtype=torch.float32, device=dev0)
                    x1, h1, w1 = self.conv_bn(blk["c1"], blk["b1"], cur, N, h, w, True, None, train, br, apply=False,
                                              scale_shift=(sc1, sh1))
                    y1 = torch.empty_like(x1)
                    mk1 = torch.empty(x1.numel() // (16 // ops._sz(self.dtype)), dtype=torch.uint8, device=dev0)
                    br[0].y, br[0].mask = y1, mk1
                    bn_in = (x1, sc1, sh1, mk1)
            if bn_in is None:
                y1, h1, w1 = self.conv_bn(blk["c1"], blk["b1"], cur, N, h, w, True, None, train, br)
            pp = blk["c3"].Ci
            cs = crow = None
            if lin:
                d2 = blk["c2"].fwd_desc(self.dtype, N, h1, w1)
                crow = ops.bn_act_colsum_rows(self.dtype, N * d2.Ho * d2.Wo // Vt, pp)
                cs = self._work("linbn_cs", Vt * crow * pp)
            y2, h2, w2 = self.conv_bn(blk["c2"], blk["b2"], y1, N, h1, w1, True, None, train, br, colsum=cs, bn_in=bn_in)
            if lin:
                slabs, cap = self._slab_buf(pp * pp, Vt)
                ns = ops.conv_wgrad_slabs(self._lin_conv_desc(self.dtype, N, h2, w2, pp, pp), y2, y2, slabs, views=Vt, cap=cap)
                br[1].gram = torch.empty(Vt * pp * pp, dtype=torch.float32, device=dev0)
                br[1].colsum = torch.empty(Vt * pp, dtype=torch.float64, device=dev0)
                ops.linbn_moments(slabs, ns, pp * pp, br[1].gram, views=Vt, colsum=cs, colsum_rows=crow,
                                  s_out=br[1].colsum, p=pp)
            ra = None
            pend = None
            # the join by linearity needs what its backward needs (conv3_backward_linbn, lin_d)
            join_lin = (lin and "cd" in blk and self.linbn_fwd and self.linbn_ds and self.linbn_join and not self._ordered_bn
                        and self.fuse_bn_bwd and blk["cd"].Ci % 64 == 0 and
                        (blk["cd"].stride == 1 or (bi > 0 and blk["cd"].stride == 2 and
                                                   (Vt == 1 or (N * h * w) % 256 == 0))))
            if join_lin:
                y3, h3, w3 = self.join_fused(blk, br[1], y2, cur, N, h, w, h2, w2, br)
                block_recs.append(br)
                cur, h, w = y3, h3, w3
                continue
            if "cd" in blk and lazy:
                # downsample branch: convolution + statistics only; its BatchNorm is applied inside the join below
                # (data parallel: its statistics travel in conv3's all-reduce)
                ra = (self._work("scale_d", 2 * 2048), self._work("shift_d", 2 * 2048))
                pend = [] if (train and self.stat_sync is not None) else None
                idn, _, _ = self.conv_bn(blk["cd"], blk["bd"], cur, N, h, w, False, None, train, br, apply=False,
                                         scale_shift=ra, pending=pend)
            elif "cd" in blk:
                idn, _, _ = self.conv_bn(blk["cd"], blk["bd"], cur, N, h, w, False, None, train, br)
            else:
                idn = cur
            if lin and "cd" not in blk and self.linbn_fwd and not self._ordered_bn:
                # conv3 -> bn3 -> (+identity) -> ReLU in ONE launch: bn3's batch statistics come from the moments of y2
                y3, h3, w3 = self.conv3_bn3_fused(blk["c3"], blk["b3"], br[1], y2, idn, N, h2, w2, br)
            else:
                y3, h3, w3 = self.conv_bn(blk["c3"], blk["b3"], y2, N, h2, w2, True, idn, train, br, res_affine=ra,
                                          pending=pend)
                if lin:
                    br[-1].linbn = True  # backward of conv3 -> bn3 by linearity; needs br[1].colsum / .gram (moments of y2)
            block_recs.append(br)
            cur, h, w = y3, h3, w3
        ops.avgpool_fwd(self.dtype, cur, feat_f32, feat_t, N, h * w, plan.out_dim)
        if save is not None:
            save.append({"plan": plan, "stem": recs[0], "stem_hw": (Ho, Wo), "pool_hw": (Hp, Wp), "argmax": amax,
                         "blocks": block_recs, "N": N, "last_hw": (h, w), "V": self._V})

    def encoder_backward(self, ctx, dfeat, last_view=True):
        """dfeat: [N,2048] `dtype` gradient of the pooled features.  On the last view of a step each stage's
        parameter gradients are final once its blocks are done: grad_ready fires per stage so the caller can
        start that bucket's all-reduce while earlier stages are still computing."""
        plan, N = ctx["plan"], ctx["N"]
        h, w = ctx["last_hw"]
        dcur = torch.empty(N * h * w, plan.out_dim, dtype=self.tdt, device=dfeat.device)
        ops.avgpool_bwd(self.dtype, dfeat, dcur, N, h * w, plan.out_dim)
        fr = None  # rows of fused BN-backward partials that came with dcur
        for bi in range(len(plan.blocks) - 1, -1, -1):
            blk, br = plan.blocks[bi], ctx["blocks"][bi]
            if "cd" in blk:
                r1, r2, rd, r3 = br
            else:
                (r1, r2, r3), rd = br, None
            if r3.linbn:
                # conv3 -> bn3 by linearity: dz (dcur, masked) feeds the weight- and data-gradient GEMMs as it is
                if fr is None:  # last block: dcur is the un-masked gradient from the pooling layer
                    C3, V3 = r3.cu.Co, r3.V
                    rows3 = r3.N * r3.Ho * r3.Wo // V3
                    prow = ops.bn_bwd_partial_rows(rows3, C3)
                    bpart = self._work("partials", V3 * prow * 2 * C3)
                    ops.bn_bwd_reduce(self.dtype, dcur, None, r3.xo, r3.mean, r3.invstd, dcur, rows3, C3, bpart,
                                      mask=r3.mask, views=V3)
                else:
                    prow, bpart = fr, self._ws[(self._lane, "fz_partials")]
                prev_r3 = ctx["blocks"][bi - 1][-1] if bi > 0 else None
                lin_d = rd is not None and (rd.linbn or (
                    self.linbn_ds and rd.cu.Ci % 64 == 0 and not rd.frozen_stats and
                    (rd.cu.stride == 1 or (prev_r3 is not None and self.fuse_bn_bwd and rd.cu.stride == 2 and
                                           (prev_r3.V == 1 or (r1.N * r1.H * r1.W) % 256 == 0)))))
                dy2, fr2, dxd = self.conv3_backward_linbn(r3, r2, dcur, bpart, prow, rd=rd, lin_d=lin_d)
                dz = None if rd is not None else dcur
            else:
                if rd is not None:
                    dx3, dxd = self.bn_backward_join(r3, rd, dcur, fused_rows=fr)
                    dz = None
                else:
                    dx3, dz = self.bn_backward(r3, dcur, keep_dz=True, fused_rows=fr)
                dy2, fr2 = self.conv_backward(r3, dx3, fuse=r2)
                del dx3
            dx2, _ = self.bn_backward(r2, dy2, keep_dz=False, fused_rows=fr2)
            dy1, fr1 = self.conv_backward(r2, dx2, fuse=r1)
            del dx2, dy2
            dx1, _ = self.bn_backward(r1, dy1, keep_dz=False, fused_rows=fr1)
            if rd is not None and isinstance(dxd, dict):
                # the downsample unit went by linearity: its data gradient is the two-segment product of dz (= dcur) and the
                # compact block input, joined with conv1's data gradient as the two-pass form's is
                prev_r3 = ctx["blocks"][bi - 1][-1] if bi > 0 else None
                cd = rd.cu
                Vd = r3.V
                Cd, Cin = cd.Co, cd.Ci
                dd = cd.compact_dgrad_desc(self.dtype, rd.N, rd.Ho, rd.Wo)
                if cd.stride == 2:
                    dsp = torch.empty(rd.N * rd.Ho * rd.Wo, Cin, dtype=self.tdt, device=dcur.device)
                    ops.conv_gemm_seg(dd, dcur, dxd["wa"], dxd["x1"], dxd["hn"], dxd["const"], dsp, None, views=Vd,
                                      w_view_stride=Cin * Cd, w1_view_stride=Cin * Cin)
                    din, fr = self.conv_backward(r1, dx1, addend=dsp, fuse=prev_r3, addend_sparse=(rd.Ho, rd.Wo))
                else:
                    din, _ = self.conv_backward(r1, dx1)
                    ops.conv_gemm_seg(dd, dcur, dxd["wa"], dxd["x1"], dxd["hn"], dxd["const"], din, din, views=Vd,
                                      w_view_stride=Cin * Cd, w1_view_stride=Cin * Cin)
                    fr = None
            elif rd is not None:
                prev_r3 = ctx["blocks"][bi - 1][-1] if bi > 0 else None
                cd = rd.cu
                V = prev_r3.V if prev_r3 is not None else 1
                if (prev_r3 is not None and self.fuse_bn_bwd and cd.stride == 2
                        and (V == 1 or (r1.N * r1.H * r1.W) % 256 == 0)):
                    # Join of the two data gradients of a stride-2 downsample block WITHOUT a second pass over the
                    # block-input gradient: the downsample convolution's data gradient is computed first, compact (it
                    # only exists at the even pixels), and conv1's data gradient takes it as a sparse addend -- so that
                    # launch sees the complete gradient of the previous block's output and runs that block's
                    # BatchNorm-backward phase 1 in its epilogue, like every other block boundary.
                    self._wgrad(cd, rd, dxd)
                    hs, ws = rd.Ho, rd.Wo
                    dd = cd.compact_dgrad_desc(self.dtype, rd.N, hs, ws)
                    dsp = torch.empty(rd.N * hs * ws, cd.Ci, dtype=self.tdt, device=dxd.device)
                    ops.conv_gemm(dd, dxd, cd.w_dgrad, dsp, None, None)
                    din, fr = self.conv_backward(r1, dx1, addend=dsp, fuse=prev_r3, addend_sparse=(hs, ws))
                else:
                    din, _ = self.conv_backward(r1, dx1)
                    self.conv_backward(rd, dxd, into=din)
                    fr = None
            else:
                # din is the gradient of the previous block's output = of its bn3 (+residual, ReLU) unit
                prev_r3 = ctx["blocks"][bi - 1][-1] if bi > 0 else None
                din, fr = self.conv_backward(r1, dx1, addend=dz, fuse=prev_r3)
            dcur = din
            if last_view and "cd" in blk and bi > 0:  # first block of a stage: the stage is complete
                stage = blk["c1"].name.rsplit(".", 2)[0]  # e.g. derm_backbone.encoder.layer4
                self._notify(stage + ".", stage + ".")
        # maxpool -> stem BN/ReLU -> stem weight gradient (no data gradient: the image needs none)
        Ho, Wo = ctx["stem_hw"]
        rs = ctx["stem"]
        # maxpool gradient gather + recomputed ReLU mask + BatchNorm-backward phase 1 in one pass
        dz = torch.empty(N * Ho * Wo, 64, dtype=self.tdt, device=dfeat.device)
        prow = ops.maxpool_bn_bwd_partial_rows(N, Ho, Wo, rs.V)
        part = self._work("fz_partials", rs.V * prow * 2 * 64)
        ops.maxpool_bn_bwd(self.dtype, ctx["argmax"], dcur, rs.xo, rs.scale, rs.shift, rs.mean, rs.invstd, dz, part,
                           N, Ho, Wo, 64, views=rs.V)
        if self.direct_stem:
            # BatchNorm-backward apply inside the stem weight gradient's operand load: d(conv1 output) never reaches HBM
            lsums, gsums, count = self._bn_backward_sums(rs, part, prow)
            bn = rs.bu.name
            wg = ops.stem_wgrad_bn16 if isinstance(rs.x_in, ops.StemImage) else ops.stem_wgrad_bn
            wg(self.dtype, rs.x_in, dz, rs.xo, rs.mean, rs.invstd, self._p(bn + ".weight"), gsums, count,
               lsums, self._g(bn + ".weight"), self._g(bn + ".bias"), self._g(rs.cu.name + ".weight"), views=rs.V,
               slabs=self._work("stem_slabs", ops.STEM_WGRAD_SLABS * 64 * 147) if self.det_wgrad else None)
        else:
            dxo, _ = self.bn_backward(rs, dz, keep_dz=False, fused_rows=prow)
            self.conv_backward(rs, dxo, need_dx=False)
        if last_view:
            self._notify(plan.prefix + "conv1", plan.prefix + "layer1.")

    # ---- projector -----------------------------------------------------------------------
    def projector_forward(self, plan, x_t, M, train, z_out, save=None):
        """x_t [M,2048] `dtype` -> z_out [M,proj_dim] fp32 (view into the caller's [2B,proj] buffer)."""
        recs = [] if save is not None else None
        h, _, _ = self.conv_bn(plan.l0, plan.b1, x_t, M, 1, 1, True, None, train, recs)
        h, _, _ = self.conv_bn(plan.l3, plan.b4, h, M, 1, 1, True, None, train, recs)
        self.conv_bn(plan.l6, plan.b7, h, M, 1, 1, False, None, train, recs, out_f32=True, y_out=z_out)
        if save is not None:
            save.append(recs)

    def projector_backward(self, recs, dz, addend=None, into=None):
        """dz [M,proj_dim] `dtype` -> gradient w.r.t. the projector input [M,2048] (+addend)."""
        r0, r3, r6 = recs
        dx, _ = self.bn_backward(r6, dz, keep_dz=False)
        d, fr = self.conv_backward(r6, dx, fuse=r3)
        dx, _ = self.bn_backward(r3, d, keep_dz=False, fused_rows=fr)
        d, fr = self.conv_backward(r3, dx, fuse=r0)
        dx, _ = self.bn_backward(r0, d, keep_dz=False, fused_rows=fr)
        if into is not None:
            return self.conv_backward(r0, dx, into=into)[0]
        return self.conv_backward(r0, dx, addend=addend)[0]

    # ---- whole model ---------------------------------------------------------------------
    @staticmethod
    def cross_pairs(style):
        return {0: [(0, 0), (1, 1)], 1: [(0, 1), (1, 0)], 2: [(0, 0), (0, 1), (1, 0), (1, 1)]}[style]

    def forward(self, views, style=0, train=True, want_grad=True, metadata=None):
        """views: dict branch -> [x_view0, x_view1] (NCHW fp32).  SimCLRSkinV32.forward / SimCLR.forward.
        Returns (zs, feats, saved): zs[name] is the fp32 [2B,proj] projector output whose NT-Xent logits the
        caller emits; feats[branch] = (fp32, dtype) pooled features [2B,2048]; saved feeds backward()."""
        first = next(iter(views.values()))[0]
        dev = first.device
        self.prepare(dev)
        self.refresh_weights(defer_lanes=self._lane_streams(dev) is not None and self.lane_cross)
        B = first.shape[0]
        saved = {"B": B, "style": style} if want_grad else None
        sv = (lambda: []) if want_grad else (lambda: None)
        zs, feats = OrderedDict(), {}
        streams = self._lane_streams(dev)
        # Cross-modal projectors (simclr.py:290-322): cross_proj[0] only ever sees dermoscopy features and cross_proj[1]
        # clinical ones, so each runs at the end of its modality's lane, hidden behind the other lane's encoder, instead of
        # on the main stream after the join (4 projector passes = ~50 small dependent launches in series).
        lane_cross = (self.cross is not None and streams is not None and set(self.branches) == {"derm", "clinic"}
                      and not self.view_lanes and self.lane_cross)
        pairs = self.cross_pairs(style) if self.cross is not None else []
        zc = [self._share(torch.empty(2 * B, self.module.proj_dim, dtype=torch.float32, device=dev), streams) for _ in pairs] \
            if lane_cross else []
        cross_recs = {}
        for key, (plan, proj) in self.branches.items():
            imgs = views[key]
            with self.lane(key, streams):
                self.prep_lane(key)  # this branch's filter banks (deferred by refresh_weights), overlapping the other lane's
            for im in imgs:
                self._share(im, streams)
            f32 = self._share(torch.empty(2 * B, plan.out_dim, dtype=torch.float32, device=dev), streams)
            ft = self._share(torch.empty(2 * B, plan.out_dim, dtype=self.tdt, device=dev), streams)
            ctxs = [None, None] if want_grad else None
            split = bool(streams) and self.view_lanes
            self._ordered_bn = split and train
            # the two views go through the encoder separately: BN statistics per view (simclr.py:58-59)
            pair = (self.pair_views and train and not split and len(imgs) == 2 and imgs[0].shape == imgs[1].shape
                    and self.pair_ok(B, imgs[0].shape[2], imgs[0].shape[3])
                    # the kernels address a tensor with 32-bit buffer offsets below 3 GB; the largest one is the stem's
                    # im2col matrix (SM3_DIRECT_STEM=0) or the stem / layer1 maps (direct stem)
                    and 2 * B * ((imgs[0].shape[2] - 1) // 2 + 1) * ((imgs[0].shape[3] - 1) // 2 + 1)
                    * (64 if self.direct_stem else STEM_KPAD) * ops._sz(self.dtype) < 0xC0000000)
            if pair:  # both views as one batch of 2B images (BatchNorm statistics still per view)
                with self.lane(key, streams):
                    tmp = [] if want_grad else None
                    self.encoder_forward(plan, [imgs[0], imgs[1]], train, f32, ft, tmp, views=2)
                    if want_grad:
                        ctxs = [tmp[0]]
            for v in (() if pair else (0, 1)):
                with self.lane(key + "#1" if (split and v == 1) else key, streams):
                    tmp = [] if want_grad else None
                    self.encoder_forward(plan, imgs[v], train, f32[v * B:(v + 1) * B], ft[v * B:(v + 1) * B], tmp)
                    if want_grad:
                        ctxs[v] = tmp[0]
            self._ordered_bn = False
            feats[key] = (f32, ft)
            precs = sv()
            with self.lane(key, streams):
                if split:
                    torch.cuda.current_stream().wait_stream(streams[key + "#1"])
                if proj is not None:  # in-modal projector on cat([f1, f2])  (simclr.py:61)
                    z = torch.empty(2 * B, self.module.proj_dim, dtype=torch.float32, device=dev)
                    self.projector_forward(proj, ft, 2 * B, train, z, precs)
                    zs[key] = self._share(z, streams)
                if lane_cross:
                    side = 0 if key == "derm" else 1
                    for ci, ab in enumerate(pairs):
                        rec = sv()
                        self.projector_forward(self.cross[side], ft[ab[side] * B:(ab[side] + 1) * B], B, train,
                                               zc[ci][side * B:(side + 1) * B], rec)
                        cross_recs[(ci, side)] = rec
            if want_grad:
                saved[key] = {"enc": ctxs, "proj": precs[0] if proj is not None else None}
        self._join(streams)
        for k in list(self.__dict__.get("_lane_prep", {})):
            self.prep_lane(k)  # (a branch whose lane never ran)
        cross_saved = []
        if lane_cross:
            for ci, (a, b) in enumerate(pairs):
                zs[f"cross{ci}"] = zc[ci]
                if want_grad:
                    cross_saved.append((a, b, cross_recs[(ci, 0)][0], cross_recs[(ci, 1)][0]))
        elif self.cross is not None:  # cross-modal: each projector sees its own B rows (simclr.py:293)
            for ci, (a, b) in enumerate(self.cross_pairs(style)):
                z = torch.empty(2 * B, self.module.proj_dim, dtype=torch.float32, device=dev)
                pa, pb = sv(), sv()
                self.projector_forward(self.cross[0], feats["derm"][1][a * B:(a + 1) * B], B, train, z[:B], pa)
                self.projector_forward(self.cross[1], feats["clinic"][1][b * B:(b + 1) * B], B, train, z[B:], pb)
                zs[f"cross{ci}"] = z
                if want_grad:
                    cross_saved.append((a, b, pa[0], pb[0]))
        if want_grad:
            saved["cross"] = cross_saved
        if metadata is not None:
            if self.meta is None:
                raise ValueError("the model was built without metadata_dim")
            pad = self.meta.l0.Ci
            if metadata.dim() != 2 or metadata.shape[0] != B or metadata.shape[1] > pad:
                raise ValueError(f"metadata must be [B, <= {pad}]")
            x32 = torch.zeros(B, pad, dtype=torch.float32, device=dev)
            x32[:, : metadata.shape[1]].copy_(metadata)
            if self.tdt == torch.float32:
                xt = x32
            else:
                xt = torch.empty(B, pad, dtype=self.tdt, device=dev)
                ops.cast_from_f32(self.dtype, x32, xt)
            zm = torch.empty(B, self.module.proj_dim, dtype=torch.float32, device=dev)
            pm = sv()
            self.projector_forward(self.meta, xt, B, train, zm, pm)
            zs["meta"] = zm  # [B, proj]: not a loss term by itself -- the trainer pairs it with the cross projections
            if want_grad:
                saved["meta"] = pm[0]
        return zs, feats, saved

    def _notify(self, plan_prefix_first, plan_prefix_last):
        if self.grad_ready is not None:
            self._sync_side()
            self.grad_ready(plan_prefix_first, plan_prefix_last)

    def backward(self, saved, dz, dfeat=None):
        """dz: dict name -> [2B,proj] `dtype` gradient of the projector outputs; dfeat: optional dict branch ->
        [2B,2048] `dtype` gradient arriving at the pooled features directly.  Accumulates parameter gradients
        into the flat gradient buffer (self.store.flat_g)."""
        B = saved["B"]
        dfe = {}
        dev = self.store.flat_p.device
        streams = self._lane_streams(dev)
        for t in dz.values():
            self._share(t, streams)
        # mirror of forward(): each cross-modal projector's backward inside its modality's lane (no join, no main-stream pass)
        lane_cross = (self.cross is not None and streams is not None and set(self.branches) == {"derm", "clinic"}
                      and not self.view_lanes and self.lane_cross and bool(saved.get("cross")) and all(f"cross{ci}" in dz
                                                                                 for ci in range(len(saved["cross"]))))
        for key, (plan, proj) in self.branches.items():
            extra = dfeat.get(key) if dfeat is not None else None
            self._share(extra, streams)
            with self.lane(key, streams):
                if proj is not None and key in dz:
                    dfe[key] = self.projector_backward(saved[key]["proj"], dz[key], addend=extra)  # [2B,2048]
                elif extra is not None:
                    dfe[key] = extra.clone()
                else:
                    dfe[key] = torch.zeros(2 * B, plan.out_dim, dtype=self.tdt, device=dev)
                self._share(dfe[key], streams)
                if lane_cross:  # this modality's cross-projector passes, then its projector buckets are final
                    side = 0 if key == "derm" else 1
                    for ci, rec in enumerate(saved["cross"]):
                        ab = rec[side]
                        self.projector_backward(rec[2 + side], dz[f"cross{ci}"][side * B:(side + 1) * B],
                                                into=dfe[key][ab * B:(ab + 1) * B])
                    if proj is not None:
                        self._notify(proj.prefix, proj.prefix)
                    self._notify(self.cross[side].prefix, self.cross[side].prefix)
                self._sync_side()
        if not lane_cross:
            self._join(streams)
            for ci, (a, b, pa, pb) in enumerate(saved["cross"]):
                d = dz[f"cross{ci}"]
                self.projector_backward(pa, d[:B], into=dfe["derm"][a * B:(a + 1) * B])
                self.projector_backward(pb, d[B:], into=dfe["clinic"][b * B:(b + 1) * B])
            # all projector gradients are final here
            for key, (plan, proj) in self.branches.items():
                if proj is not None:
                    self._notify(proj.prefix, proj.prefix)
            if self.cross is not None:
                self._notify(self.cross[0].prefix, self.cross[-1].prefix)
        if self.meta is not None:
            if "meta" in dz and saved.get("meta") is not None:
                self.projector_backward(saved["meta"], dz["meta"])
            self._notify(self.meta.prefix, self.meta.prefix)
        split = bool(streams) and self.view_lanes
        for key, (plan, proj) in self.branches.items():
            if len(saved[key]["enc"]) == 1:  # both views went through as one batch
                with self.lane(key, streams):
                    self.encoder_backward(saved[key]["enc"][0], dfe[key], last_view=True)
                    saved[key]["enc"][0] = None
                    self._sync_side()
                continue
            for v in (1, 0):
                with self.lane(key + "#1" if (split and v == 1) else key, streams):
                    # with the views on two lanes a stage's gradients are final only when BOTH are done: the
                    # per-stage notifications of the last view are replaced by one round after the join below
                    self.encoder_backward(saved[key]["enc"][v], dfe[key][v * B:(v + 1) * B],
                                          last_view=(v == 0 and not split))
                    saved[key]["enc"][v] = None  # free the view's activations as soon as it is done
                    self._sync_side()
            if split:
                with self.lane(key, streams):
                    torch.cuda.current_stream().wait_stream(streams[key + "#1"])
                    for li in (4, 3, 2):
                        self._notify(f"{plan.prefix}layer{li}.", f"{plan.prefix}layer{li}.")
                    self._notify(plan.prefix + "conv1", plan.prefix + "layer1.")
        self._sync_side()
        self._join(streams)

    def encoder_only(self, branch, x, train, want_grad):
        """One encoder call (SimCLRSkinV3.extract / a bare ResNet forward): fp32 features [N,2048] and the
        context for encoder_backward when want_grad."""
        dev = x.device
        self.prepare(dev)
        self.refresh_weights()
        plan = self.branches[branch][0]
        N = x.shape[0]
        f32 = torch.empty(N, plan.out_dim, dtype=torch.float32, device=dev)
        ctxs = [] if want_grad else None
        self.encoder_forward(plan, x, train, f32, None, ctxs)
        return f32, (ctxs[0] if want_grad else None)
